// fp32 parity mode, 3x3x3 convolutions on the bf16 matrix cores through exact three-limb operand splitting (igemm_k3x.h)
#include <stdlib.h>
#include "igemm_dispatch.h"
#include "igemm_k3x.h"

// ck = min(C, 16); mt = 16 or 32 rows per workgroup
int g1_dispatch_k3_x3(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    // 8 stored input and output channels (the full-resolution layers; out_block's two logits are stored in 8): Toeplitz rows, weights packed to match
    if (ck == 8 && p.C == 8 && p.M == 8 && p.nch == 1 && vs_k3x_toeplitz(8, 8, 27))
        return epi == EPI_SOFTMAX2 ? k3xt_launch<EPI_SOFTMAX2>(p, s) : k3xt_launch<EPI_RAW>(p, s);
    // fused apply: k3xt_kernel, and k3x_kernel<8, 16, RAW, .., MULTI, FA> for 16 stored input channels (vs_conv_k3_fused_apply_supported says so beforehand)
    if (p.fa_x != nullptr && !(ck == 8 && mt == 16 && epi == EPI_RAW && p.nch == 2 && p.C == 16)) return VS_ESHAPE;
    if (epi == EPI_SOFTMAX2) {
        if (ck != 8 || mt != 16 || p.nch != 1) return VS_ESHAPE;      // out_block: 8 stored channels -> 2 logits
        return k3x_launch<8, 16, EPI_SOFTMAX2, false>(p, tiles, row_tiles, s);
    }
    if (ck == 8) {
        if (p.nch == 1) {
            if (mt == 16) return k3x_launch<8, 16, EPI_RAW, false>(p, tiles, row_tiles, s);
            if (mt == 32) return k3x_launch<8, 32, EPI_RAW, false>(p, tiles, row_tiles, s);
        } else {                                          // VS_K3X_CK=8: 8-channel chunks for every layer (two workgroups per CU)
            if (p.ea_sync != nullptr && mt != 16) return VS_ESHAPE;       // the epilogue apply exists for the 16-row workgroups (conv_api.hip asks k3x_ea_capacity first)
            if (mt == 16 && !p.fa_x && vs_cfg().k3_short_tiles) {
                // the under-filled launches of the 12^3-class levels (<= 128 workgroups): 4 x 2 x 16 tiles, 4 x 1 x 16 where that still leaves <= 128 (igemm_k3_h16.inc)
                const long long zx = (long long)p.N * ((p.D + 3) / 4) * p.txn;
                // (4 x 2 x 16 for EVERY launch — two waves per SIMD instead of one — measured slower: fp32 step 5.748 -> 5.846 ms; the halo grows 2.5 x -> 3.4 x)
                if ((long long)tiles * row_tiles <= 128 && zx * ((p.H + 1) / 2) * row_tiles <= 256) {
                    if (vs_cfg().k3_short_tiles >= 2 && zx * ((p.H + 1) / 2) * row_tiles <= 128 && zx * p.H * row_tiles <= 256)
                        return k3x_launch_short<8, 16, true, 1>(p, tiles, row_tiles, s);
                    return k3x_launch_short<8, 16, true, 2>(p, tiles, row_tiles, s);
                }
            }
            if (mt == 16 && p.ea_sync != nullptr) return k3x_launch_short<8, 16, true, 4>(p, tiles, row_tiles, s);
            if (mt == 16) return k3x_launch<8, 16, EPI_RAW, true>(p, tiles, row_tiles, s);
            if (mt == 32) return k3x_launch<8, 32, EPI_RAW, true>(p, tiles, row_tiles, s);
        }
        return VS_ESHAPE;
    }
    if (p.ea_sync != nullptr) return VS_ESHAPE;
    if (ck == 16) {
        if (p.nch == 1) {
            if (mt == 16) return k3x_launch<16, 16, EPI_RAW, false>(p, tiles, row_tiles, s);
            if (mt == 32) return k3x_launch<16, 32, EPI_RAW, false>(p, tiles, row_tiles, s);
        } else {
            if (mt == 16) return k3x_launch<16, 16, EPI_RAW, true>(p, tiles, row_tiles, s);
            if (mt == 32) return k3x_launch<16, 32, EPI_RAW, true>(p, tiles, row_tiles, s);
        }
    }
    return VS_ESHAPE;
}

// conv_api.hip: workgroups of a k3x_kernel<8, 16, .., EA> launch that are certainly resident together
int k3x_ea_capacity(int n, int c, int m) { return k3x_ea_max_wgs(n, c, m); }
