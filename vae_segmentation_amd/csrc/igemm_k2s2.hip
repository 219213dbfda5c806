#include <stdlib.h>
#include "igemm_dispatch.h"
int g1_dispatch_k2s2_f16(const G1Params& p, int ck, int mt, int tiles, int row_tiles, hipStream_t s);
int g1_dispatch_k2s2(const G1Params& p, int dtype, int ck, int mt, int tiles, int row_tiles, hipStream_t s) {
    if (dtype == VS_F32) { if (g1_f32_limbs()) { G1E_ALL(float, G1_K2S2, EPI_RAW, true) G1L_ALL(G1_K2S2, EPI_RAW) } else { G1E_ALL(float, G1_K2S2, EPI_RAW, false) G1_ALL(float, G1_K2S2, EPI_RAW) } }
    else if (dtype == VS_BF16) { G1E_ALL(unsigned short, G1_K2S2, EPI_RAW, false) G1_ALL(unsigned short, G1_K2S2, EPI_RAW) }
    else if (dtype == VS_F16) return g1_dispatch_k2s2_f16(p, ck, mt, tiles, row_tiles, s);
    return VS_ESHAPE;
}
