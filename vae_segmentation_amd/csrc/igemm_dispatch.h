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
