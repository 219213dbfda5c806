// fp16 instantiations of g1_kernel (stride-2 and transposed convolutions); own translation unit for build parallelism
#include "igemm_dispatch.h"
int g1_dispatch_k2s2_f16(const G1Params& p, int ck, int mt, int tiles, int row_tiles, hipStream_t s) {
    G1E_ALL(vs_half, G1_K2S2, EPI_RAW, false)
    G1_ALL(vs_half, G1_K2S2, EPI_RAW)
    return VS_ESHAPE;
}
int g1_dispatch_pw_f16(const G1Params& p, int ck, int mt, int tiles, int row_tiles, hipStream_t s) {
    G1E_ALL(vs_half, G1_PW, EPI_SCATTER, false)
    G1_ALL(vs_half, G1_PW, EPI_SCATTER)
    return VS_ESHAPE;
}
