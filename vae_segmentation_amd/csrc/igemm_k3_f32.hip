#include <stdlib.h>
#include "igemm_dispatch.h"
#include "igemm_k3.h"
#include "igemm_k3s.h"

#define K3_CASE(CKV, MTV) if (ck == CKV && mt == MTV) return k3_launch<float, CKV, MTV, EPI_RAW>(p, tiles, row_tiles, s);
#define K3_ALL_MT(CKV) K3_CASE(CKV, 16) K3_CASE(CKV, 32) K3_CASE(CKV, 64)

int g1_dispatch_k3_f32(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    if (ck == 8 && p.C == 8 && p.M == 8 && mt == 16) {     // the 8-channel full-resolution layers: y-Toeplitz rows (weights packed to match: vs_k3_toeplitz_f32)
        if (epi == EPI_SOFTMAX2) return k3_launch<float, 8, 16, EPI_SOFTMAX2, 4, true>(p, tiles, row_tiles, s);
        return k3_launch<float, 8, 16, EPI_RAW, 4, true>(p, tiles, row_tiles, s);
    }
    if (epi == EPI_SOFTMAX2) return VS_ESHAPE;             // out_block is an 8-channel layer (above)
    if (k3s_takes(p, ck))                                 // volumes up to 6^3: flattened columns, the waves split the taps (igemm_k3s.h)
        return p.sums ? k3s_launch<float, true>(p, s) : k3s_launch<float, false>(p, s);
    // C >= 32 (the 24^3 level and below): the exact-f32 MFMA makes these layers MFMA-cycle bound per wave, and 4x4x16 tiles of 16-32 rows leave
    // most SIMDs idle (12^3 x 64: 72 workgroups).  16-row blocks and the tallest tile (4, 2 or 1 rows of 16 voxels per wave) that still
    // gives >= VS_K3F_MIN_WGS workgroups (default 512 = two waves per SIMD).
    if (ck == 32) {
        const int min_wgs = vs_cfg().k3f_min_wgs;
        const long long rows16 = (long long)row_tiles * (mt / 16);
        const long long zx = (long long)((p.D + 3) / 4) * p.txn * p.N;
        if (min_wgs > 0 && zx * ((p.H + 3) / 4) * rows16 < min_wgs) {
            if (zx * ((p.H + 1) / 2) * rows16 >= min_wgs) return k3_launch<float, 32, 16, EPI_RAW, 2>(p, tiles, (int)rows16, s);
            return k3_launch<float, 32, 16, EPI_RAW, 1>(p, tiles, (int)rows16, s);
        }
    }
    K3_ALL_MT(8) K3_ALL_MT(16) K3_ALL_MT(32)
    return VS_ESHAPE;
}

// chain.h: the DoubleConv chains of the small volumes, fp32 storage (exact-f32 MFMA, as k3s_kernel<float>)
int chain_dispatch_k3s_f32(const K3Chain& c, int bwd, hipStream_t s) { return k3s_chain_launch<float>(c, bwd != 0, s); }
