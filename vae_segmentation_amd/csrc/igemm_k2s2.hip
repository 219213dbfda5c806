#include "igemm_dispatch.h"
int g1_dispatch_k2s2(const G1Params& p, int dtype, int ck, int mt, int tiles, int row_tiles, hipStream_t s) {
    if (dtype == VS_F32) { G1_ALL(float, G1_K2S2, EPI_RAW) }
    else { G1_ALL(unsigned short, G1_K2S2, EPI_RAW) }
    return VS_ESHAPE;
}
