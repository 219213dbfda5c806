// fp16 instantiations of the 16-bit 3x3x3 kernels (own translation unit: they compile in parallel with the bf16 ones)
#undef VS_STAMPS
#undef VS_STAMPS_LITE
#include "igemm_k3_h16.inc"

int g1_dispatch_k3_f16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    return dispatch_k3_h16<vs_half>(p, ck, mt, epi, tiles, row_tiles, s);
}

int chain_dispatch_k3s_f16(const K3Chain& c, int bwd, hipStream_t s) { return k3s_chain_launch<vs_half>(c, bwd != 0, s); }
