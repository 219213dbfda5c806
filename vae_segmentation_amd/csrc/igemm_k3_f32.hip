#include "igemm_dispatch.h"
#include "igemm_k3.h"

#define K3_CASE(CKV, MTV) if (ck == CKV && mt == MTV) return k3_launch<float, CKV, MTV, EPI_RAW>(p, tiles, row_tiles, s);
#define K3_ALL_MT(CKV) K3_CASE(CKV, 16) K3_CASE(CKV, 32) K3_CASE(CKV, 64)

int g1_dispatch_k3_f32(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    if (epi == EPI_SOFTMAX2) {
        if (ck == 8 && mt == 16) return k3_launch<float, 8, 16, EPI_SOFTMAX2>(p, tiles, row_tiles, s);
        return VS_ESHAPE;
    }
    K3_ALL_MT(8) K3_ALL_MT(16) K3_ALL_MT(32)
    return VS_ESHAPE;
}
