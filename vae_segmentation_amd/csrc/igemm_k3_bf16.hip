#include "igemm_dispatch.h"
int g1_dispatch_k3_bf16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    if (epi == EPI_SOFTMAX2) {
        if (ck == 8 && mt == 16) return g1_launch<unsigned short, 8, G1_K3, 16, EPI_SOFTMAX2>(p, tiles, row_tiles, s);
        return VS_ESHAPE;
    }
    G1_ALL(unsigned short, G1_K3, EPI_RAW)
    return VS_ESHAPE;
}
