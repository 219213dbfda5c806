#include <stdlib.h>
#include "igemm_dispatch.h"
int g1_dispatch_pw_f16(const G1Params& p, int ck, int mt, int tiles, int row_tiles, hipStream_t s);
int g1_dispatch_pw(const G1Params& p, int dtype, int ck, int mt, int tiles, int row_tiles, hipStream_t s) {
    if (dtype == VS_F32) { if (g1_f32_limbs()) { G1E_ALL(float, G1_PW, EPI_SCATTER, true) G1L_ALL(G1_PW, EPI_SCATTER) } else { G1E_ALL(float, G1_PW, EPI_SCATTER, false) G1_ALL(float, G1_PW, EPI_SCATTER) } }
    else if (dtype == VS_BF16) { G1E_ALL(unsigned short, G1_PW, EPI_SCATTER, false) G1_ALL(unsigned short, G1_PW, EPI_SCATTER) }
    else if (dtype == VS_F16) return g1_dispatch_pw_f16(p, ck, mt, tiles, row_tiles, s);
    return VS_ESHAPE;
}
