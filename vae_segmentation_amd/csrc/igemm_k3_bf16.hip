// bf16 instantiations of the 16-bit 3x3x3 kernels
#define VS_CHAIN_STAMPS_TU 1    // the diagnostic chain stamps (-DVS_CHAIN_STAMPS) live in this translation unit only
#include "igemm_k3_h16.inc"

int g1_dispatch_k3_bf16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    return dispatch_k3_h16<unsigned short>(p, ck, mt, epi, tiles, row_tiles, s);
}

// conv_api.hip: is there a fused-apply kernel for this (already planned) backward-data launch?  The same for both 16-bit storage types.
int g1_k3_fa_supported(const G1Params& p, int ck, int mt) { return k3_h16_fa_supported(p, ck, mt) ? 1 : 0; }
int k3tw_slab_count(int n, int d, int h, int w) { return k3tw_grid(n, d, h, w); }

// chain.h: the DoubleConv chains of the small volumes
int chain_dispatch_k3s_bf16(const K3Chain& c, int bwd, hipStream_t s) { return k3s_chain_launch<unsigned short>(c, bwd != 0, s); }
int k3b_ea_capacity(int n, int c, int m) { return k3b_ea_max_wgs(n, c, m); }
