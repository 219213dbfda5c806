// bf16 instantiations of the 16-bit 3x3x3 kernels
#include "igemm_k3_h16.inc"

int g1_dispatch_k3_bf16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    return dispatch_k3_h16<unsigned short>(p, ck, mt, epi, tiles, row_tiles, s);
}
