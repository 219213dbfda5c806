// Dispatch tables for the g1_kernel instantiations; each igemm_*.hip instantiates one KIND.
#pragma once
#include "igemm.h"

int g1_dispatch_k2s2(const G1Params& p, int dtype, int ck, int mt, int tiles, int row_tiles, hipStream_t s);
int g1_dispatch_pw(const G1Params& p, int dtype, int ck, int mt, int tiles, int row_tiles, hipStream_t s);

#define G1_CASE(T, CKV, KIND, MTV, EPI) \
    if (ck == CKV && mt == MTV) return g1_launch<T, CKV, KIND, MTV, EPI>(p, tiles, row_tiles, s);

#define G1_ALL_MT(T, CKV, KIND, EPI) \
    G1_CASE(T, CKV, KIND, 16, EPI) G1_CASE(T, CKV, KIND, 32, EPI) G1_CASE(T, CKV, KIND, 64, EPI)

#define G1_ALL(T, KIND, EPI) G1_ALL_MT(T, 8, KIND, EPI) G1_ALL_MT(T, 16, KIND, EPI) G1_ALL_MT(T, 32, KIND, EPI)
// fp32 parity mode: the limb form of g1_kernel (igemm.h, LIMB) unless VS_F32_LIMBS=0 asks for the exact-f32 MFMA everywhere
#define G1L_CASE(CKV, KIND, MTV, EPI) \
    if (ck == CKV && mt == MTV) return g1_launch<float, CKV, KIND, MTV, EPI, true>(p, tiles, row_tiles, s);
#define G1L_ALL_MT(CKV, KIND, EPI) G1L_CASE(CKV, KIND, 16, EPI) G1L_CASE(CKV, KIND, 32, EPI) G1L_CASE(CKV, KIND, 64, EPI)
#define G1L_ALL(KIND, EPI) G1L_ALL_MT(8, KIND, EPI) G1L_ALL_MT(16, KIND, EPI) G1L_ALL_MT(32, KIND, EPI)
// the epilogue-apply instantiations (p.ea_sync given): 16-row workgroups, every channel chunk width
#define G1E_ALL(T, KIND, EPI, LIMBV) \
    if (p.ea_sync != nullptr) { \
        if (mt == 16 && ck == 8) return g1_launch<T, 8, KIND, 16, EPI, LIMBV, true>(p, tiles, row_tiles, s); \
        if (mt == 16 && ck == 16) return g1_launch<T, 16, KIND, 16, EPI, LIMBV, true>(p, tiles, row_tiles, s); \
        if (mt == 16 && ck == 32) return g1_launch<T, 32, KIND, 16, EPI, LIMBV, true>(p, tiles, row_tiles, s); \
        return VS_ESHAPE; \
    }
static inline bool g1_f32_limbs() {
    const int on = vs_cfg().f32_limbs;
    const int g1 = vs_cfg().g1_limbs;              // A/B switch of this kernel family alone
    return on != 0 && g1 != 0;
}
