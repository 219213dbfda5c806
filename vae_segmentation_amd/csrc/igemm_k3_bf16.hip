#include "igemm_dispatch.h"
#include "igemm_k3b.h"
#include "igemm_k3t.h"
#include "igemm_k3s.h"

#define K3B_CASE(CKV, MTV)                                                                              \
    if (ck == CKV && mt == MTV)                                                                          \
        return p.sums ? k3b_launch<CKV, MTV, EPI_RAW, true>(p, tiles, row_tiles, s)                      \
                      : k3b_launch<CKV, MTV, EPI_RAW, false>(p, tiles, row_tiles, s);

// bf16 3x3x3 convolutions run k3b_kernel with 16- or 32-row tiles (a 64-row weight block does not fit LDS next to the halo
// tile): a 64-row request from pick_mt() is served as twice as many 32-row workgroups.
int g1_dispatch_k3_bf16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    if (ck == 8 && p.C == 8 && p.M == 8) {                // the 8-channel full-resolution layers: Toeplitz kernel (weights packed to match)
        if (epi == EPI_SOFTMAX2) return k3t_launch<EPI_SOFTMAX2, false, 8>(p, s);
        return p.sums ? k3t_launch<EPI_RAW, true, 8>(p, s) : k3t_launch<EPI_RAW, false, 8>(p, s);
    }
    if (epi == EPI_RAW && k3s_takes(p, ck))                 // the small volumes of the deep levels: flattened columns, waves split the taps
        return p.sums ? k3s_launch<true>(p, s) : k3s_launch<false>(p, s);
    const bool tall = mt == 16 && ck < 32 && k3b_use_tall(p);
    if (epi == EPI_SOFTMAX2) return VS_ESHAPE;           // out_block is an 8-channel layer (above)
    if (tall) {
        if (ck == 8) return p.sums ? k3b_launch<8, 16, EPI_RAW, true, 8>(p, tiles, row_tiles, s) : k3b_launch<8, 16, EPI_RAW, false, 8>(p, tiles, row_tiles, s);
        return p.sums ? k3b_launch<16, 16, EPI_RAW, true, 8>(p, tiles, row_tiles, s) : k3b_launch<16, 16, EPI_RAW, false, 8>(p, tiles, row_tiles, s);
    }
    if (mt == 64) { mt = 32; row_tiles *= 2; }
    K3B_CASE(8, 16) K3B_CASE(8, 32) K3B_CASE(16, 16) K3B_CASE(16, 32) K3B_CASE(32, 16) K3B_CASE(32, 32)
    return VS_ESHAPE;
}
