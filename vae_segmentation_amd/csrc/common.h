// Shared device helpers for libvaeseg (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/vaeseg.h"

// the library's tuning switches (config.hip; include/vaeseg.h vs_config): read on every call, set through vs_set_config() only
const vs_config& vs_cfg();

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 vs_half;                                   // IEEE fp16 storage type (VS_F16); bf16 storage is carried as `unsigned short` bits (VS_BF16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VS_WAVE 64

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// element type traits: T = float, unsigned short (bf16 bits) or vs_half (fp16)
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int EPL = 4;   // elements per 16-byte lane fragment
    static constexpr int KG = 16;   // k's per k-group (4 lane groups x EPL)
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct ET<unsigned short> {
    static constexpr int EPL = 8;
    static constexpr int KG = 32;
    __device__ static __forceinline__ float ld(const unsigned short* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(unsigned short* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return round_bf(v); }
};

template <> struct ET<vs_half> {
    static constexpr int EPL = 8;
    static constexpr int KG = 32;
    __device__ static __forceinline__ float ld(const vs_half* p) { return (float)*p; }
    __device__ static __forceinline__ void st(vs_half* p, float v) { *p = (vs_half)v; }              // v_cvt_f16_f32: RNE
    __device__ static __forceinline__ float rnd(float v) { return (float)(vs_half)v; }
};

// VS_* dtype enum of a storage type, element size
template <typename T> struct DT;
template <> struct DT<float> { static constexpr int vs = VS_F32; };
template <> struct DT<unsigned short> { static constexpr int vs = VS_BF16; };
template <> struct DT<vs_half> { static constexpr int vs = VS_F16; };
static inline int vs_esize(int dtype) { return dtype == VS_F32 ? 4 : 2; }
static inline bool vs_dtype_ok(int dtype) { return dtype == VS_F32 || dtype == VS_BF16 || dtype == VS_F16; }

// unpack a 16-byte fragment into EPL floats / pack back
__device__ __forceinline__ void frag_unpack(const u32x4& r, float (&v)[4], float*) {
    v[0] = __uint_as_float(r[0]); v[1] = __uint_as_float(r[1]);
    v[2] = __uint_as_float(r[2]); v[3] = __uint_as_float(r[3]);
}
__device__ __forceinline__ void frag_unpack(const u32x4& r, float (&v)[8], unsigned short*) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 frag_pack(const float (&v)[4], float*) {
    u32x4 r;
    r[0] = __float_as_uint(v[0]); r[1] = __float_as_uint(v[1]);
    r[2] = __float_as_uint(v[2]); r[3] = __float_as_uint(v[3]);
    return r;
}
__device__ __forceinline__ u32x4 frag_pack(const float (&v)[8], unsigned short*) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (unsigned int)f2bf(v[2 * i]) | ((unsigned int)f2bf(v[2 * i + 1]) << 16);
    return r;
}

__device__ __forceinline__ void frag_unpack(const u32x4& r, float (&v)[8], vs_half*) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int w = r[i];             // a scalar copy first: __builtin_bit_cast applied to the vector ELEMENT r[i] was compiled as a
        const f16x2 h = __builtin_bit_cast(f16x2, w);   // cast of the vector's first dword for every i (hipcc 7.2; seen in the ISA and on the GPU)
        v[2 * i] = (float)h[0];
        v[2 * i + 1] = (float)h[1];
    }
}
__device__ __forceinline__ u32x4 frag_pack(const float (&v)[8], vs_half*) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16x2 h;
        h[0] = (vs_half)v[2 * i]; h[1] = (vs_half)v[2 * i + 1];
        const unsigned int w = __builtin_bit_cast(unsigned int, h);
        r[i] = w;
    }
    return r;
}

// ---- MFMA: one 16-byte A fragment x one 16-byte B fragment -> 16x16 f32 tile ---------------------
// bf16: one v_mfma_f32_16x16x32_bf16 (k = 8*(lane>>4)+j, j = 0..7).
// f32 : four v_mfma_f32_16x16x4_f32; MFMA j covers k = 4*(lane>>4)+j, so a lane's four k's are contiguous.
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c, unsigned short*) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c, vs_half*) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c, float*) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    return c;
}

// ---- wave reductions ---------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane of the row: four DPP moves (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) instead of four ds_bpermute round trips per __shfl_xor chain
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- per-(n,c) statistics: double[VS_STAT_SLOTS][pairs][2] --------------------------------------------------------------------------
// A statistics buffer holds VS_STAT_SLOTS partial copies of its (sum, sumsq) [or IN-backward (sum g*mask, sum g*mask*xhat)] pairs.  A
// producing workgroup accumulates into copy (blockIdx.x mod VS_STAT_SLOTS); consumers add the copies (fixed order) when they build their
// tables.  Why: the accumulation is one fp64 atomic per (workgroup, channel, statistic), device-scope atomics on ONE address retire
// ~20 ns apart, and in the one-wave kernels of the 48^3 / 96^3 levels 200-400 workgroups finish together — measured 3.5-4.4 us of a
// 10-16 us launch (tools/atomics_probe.py).  More copies shorten that chain but every consuming workgroup reads all of them while it builds
// its tables: 4 copies measured best (step time with 1 / 2 / 4 / 8 copies: 2.908 / 2.820 / 2.797 / 3.021 ms).  Copies of one pair placed in
// the same cache line (VS_STAT_INTERLEAVE 1) gain nothing (2.92 ms): the atomics of one line retire one after another, whatever the address.
#ifndef VS_STAT_INTERLEAVE
#define VS_STAT_INTERLEAVE 0       // 0: double[slot][pair][2] (copies far apart)   1: double[pair][slot][2] (a pair's copies in one cache line)
#endif
__device__ __forceinline__ size_t stat_index(size_t pair, size_t pairs, int slot) {
    return VS_STAT_INTERLEAVE ? (pair * VS_STAT_SLOTS + slot) * 2 : ((size_t)slot * pairs + pair) * 2;
}

// ---- deterministic build (-DVS_DET_BUILD=1 -> libvaeseg_det.so; the package loads it for env VS_DETERMINISTIC=1 / ops.set_deterministic) -----
// fp64 atomics add in arrival order, so two runs of the same step differ in the last bits of every statistic (and a ReLU mask can flip on
// that).  In the deterministic build the SAME buffer, double[4][pairs][2], holds — instead of four partial fp64 copies — four signed 64-bit
// fixed-point LIMBS of one exact sum: a workgroup's partial `tot` is split exactly into 40-bit pieces of weight 2^40, 2^0, 2^-40, 2^-80
// (a 53-bit double spans at most three of them; what lies below 2^-80 is dropped, what lies above 2^80 does not occur) and each piece is
// added with an INTEGER atomic, which commutes: the limbs, hence every statistic, are independent of the arrival order.  23 bits of
// headroom per limb = 8 M partials.  Consumers combine the limbs in a fixed order.
// A compile-time switch, not a run-time flag: as a kernel parameter the untaken limb code cost the default mode 1 % of the 96^3 step
// (2.675 -> 2.702 ms, same box), and 5 % with the limb code kept out of line (2.69 -> 2.82); the throughput library is byte-for-byte free of it.
#ifndef VS_DET_BUILD
#define VS_DET_BUILD 0
#endif
static_assert(VS_STAT_SLOTS == 4, "deterministic statistics use the four slots as four fixed-point limbs");

__device__ __forceinline__ void stat_load(const double* __restrict__ st, size_t pair, size_t pairs, double (&out)[2]) {
#if VS_DET_BUILD
    long long a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s] = __double_as_longlong(st[stat_index(pair, pairs, s)]);
        b[s] = __double_as_longlong(st[stat_index(pair, pairs, s) + 1]);
    }
    out[0] = ((double)a[0] * 0x1p40 + (double)a[1]) + ((double)a[2] * 0x1p-40 + (double)a[3] * 0x1p-80);
    out[1] = ((double)b[0] * 0x1p40 + (double)b[1]) + ((double)b[2] * 0x1p-40 + (double)b[3] * 0x1p-80);
#else
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int s = 0; s < VS_STAT_SLOTS; ++s) {
        a += st[stat_index(pair, pairs, s)];
        b += st[stat_index(pair, pairs, s) + 1];
    }
    out[0] = a; out[1] = b;
#endif
}
// stat_load in two halves: the four copies (limbs) requested now, combined later (the same arithmetic, in the same order) — for a consumer that needs the pair only
// in its epilogue and must not wait for cold lines in its prologue (igemm_k3s.h, the mask tensor's statistics of the backward bodies)
__device__ __forceinline__ void stat_load_raw(const double* __restrict__ st, size_t pair, size_t pairs, double (&raw)[4][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        raw[s][0] = st[stat_index(pair, pairs, s)];
        raw[s][1] = st[stat_index(pair, pairs, s) + 1];
    }
}
__device__ __forceinline__ void stat_combine(const double (&raw)[4][2], double (&out)[2]) {
#if VS_DET_BUILD
    long long a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { a[s] = __double_as_longlong(raw[s][0]); b[s] = __double_as_longlong(raw[s][1]); }
    out[0] = ((double)a[0] * 0x1p40 + (double)a[1]) + ((double)a[2] * 0x1p-40 + (double)a[3] * 0x1p-80);
    out[1] = ((double)b[0] * 0x1p40 + (double)b[1]) + ((double)b[2] * 0x1p-40 + (double)b[3] * 0x1p-80);
#else
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int s = 0; s < VS_STAT_SLOTS; ++s) { a += raw[s][0]; b += raw[s][1]; }
    out[0] = a; out[1] = b;
#endif
}
// this workgroup's contribution `tot` to statistic `which` (0 / 1) of `pair`: one fp64 atomic into copy (blockIdx.x mod 4), or —
// deterministic build — four integer atomics into the four limbs
__device__ __forceinline__ void stat_add(double* st, size_t pair, size_t pairs, int which, double tot) {
#if VS_DET_BUILD
    double r = tot;
    const double up[4] = {0x1p-40, 1.0, 0x1p40, 0x1p80}, down[4] = {0x1p40, 1.0, 0x1p-40, 0x1p-80};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double q = trunc(r * up[s]);               // exact: a power-of-two scale; |q| < 2^40 while |tot| < 2^80
        r -= q * down[s];                                // exact: removes the leading bits of r
        atomicAdd((unsigned long long*)(st + stat_index(pair, pairs, s) + which), (unsigned long long)(long long)q);
    }
#else
    atomicAdd(st + stat_index(pair, pairs, (int)(blockIdx.x & (VS_STAT_SLOTS - 1))) + which, tot);
#endif
}

// mean / rstd of one (n,c) from the fp64 (sum, sumsq) pair a producer accumulated
__device__ __forceinline__ void pair_to_mean_rstd(const double (&st)[2], double inv_count, float eps, float& mean, float& rstd) {
    double m = st[0] * inv_count;
    double var = st[1] * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}
__device__ __forceinline__ void stats_to_mean_rstd(const double* st, size_t pair, size_t pairs, double inv_count, float eps, float& mean, float& rstd) {
    double v[2];
    stat_load(st, pair, pairs, v);
    pair_to_mean_rstd(v, inv_count, eps, mean, rstd);
}

// same for the 16-bit-storage kernels: the cancellation-prone part (variance) stays fp64, the reciprocal square root is v_rsq_f32 plus
// one Newton step (~1e-7 relative) instead of an fp64 sqrt + divide (~60 dependent instructions in every kernel prologue)
__device__ __forceinline__ void stats_to_mean_rstd_fast(const double* st, double inv_count, float eps, float& mean, float& rstd) {     // st: an already summed pair
    const double m = st[0] * inv_count;
    double var = st[1] * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    const float v = (float)(var + (double)eps);
    float r = rsqrtf(v);
    r = r * (1.5f - 0.5f * v * r * r);
    mean = (float)m;
    rstd = r;
}
__device__ __forceinline__ void stats_to_mean_rstd_fast(const double* st, size_t pair, size_t pairs, double inv_count, float eps, float& mean, float& rstd) {
    double v[2];
    stat_load(st, pair, pairs, v);
    stats_to_mean_rstd_fast(v, inv_count, eps, mean, rstd);
}

// counter-based Bernoulli(keep) for dropout: splitmix64 finaliser of (seed, index) -> uniform in [0,1)
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx) {
    unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx, float p) {
    return hash_uniform(seed, idx) >= p ? 1.0f / (1.0f - p) : 0.0f;
}

// ---- bounds-checked buffer access, packed normalise (shared by the bf16 conv and weight-gradient kernels) ----------
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// raw buffer load: lanes whose byte offset is >= num_records return 0 (hardware bounds check, stride 0)
__device__ i32x4 vs_raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

typedef __attribute__((ext_vector_type(2))) int i32x2;
__device__ i32x2 vs_raw_buffer_load_b64(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ int vs_raw_buffer_load_b32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
// raw buffer store: lanes whose byte offset is out of range are dropped
__device__ void vs_raw_buffer_store_b128(i32x4 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
__device__ void vs_raw_buffer_store_b64(i32x2 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");

// n / d for 0 <= n < 2^31 with the host-made pair (m, s) of k3b_fastdiv(): (mulhi(n, m) + n) >> s  (Granlund-Montgomery round-up)
__device__ __forceinline__ int fdiv(int n, unsigned int m, unsigned int sh) { return (int)((__umulhi((unsigned int)n, m) + (unsigned int)n) >> sh); }

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned int bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned int)a;
    r[1] = (int)(unsigned int)((a >> 32) & 0xffffu);     // stride 0, no swizzle
    r[2] = (int)bytes;
    r[3] = 0x00020000;                                   // gfx9 raw buffer, 32-bit data format
    return r;
}

// The two 16-bit storage formats of the throughput kernels (k3b / k3t / k3s / g3b are templated on T = unsigned short | vs_half):
// one dword = two elements; pack2 rounds to nearest even in both formats, lo / hi widen exactly.
// One element of the InstanceNorm+ReLU backward apply, rstd * (g * [xhat > 0] - m1 - xhat * m2), with its floating-point contraction PINNED (one fused
// multiply-add, written out; nothing else may fuse).  Under the default -ffp-contract=fast the optimiser fuses `gm - a - xh * b` differently from one kernel to
// the next; the standalone pass (norm.hip), the chains' in-place apply (chain.h) and every epilogue apply (igemm_k3b.h, igemm_k3x.h, igemm.h) must agree bit for
// bit in the deterministic build, so they all go through this one function.  -> the un-rounded fp32 value.
__device__ __forceinline__ float vs_in_bwd_apply1(float g, float x, float mean, float rstd, float m1, float m2) {
#pragma clang fp contract(off)
    const float xh = (x - mean) * rstd;
    const float gm = xh > 0.f ? g : 0.f;
    const float base = gm - m1;
    float out = rstd * __builtin_fmaf(-xh, m2, base);
    // the product is an fp32 VALUE before anything rounds it to 16 bits: left to itself the backend fuses `multiply, then convert` into v_fma_mixlo_f16 (one rounding of
    // the exact product) in some kernels and not in others — 29 of 884,736 outputs of a 24^3 x 32 fp16 layer differed by an ulp between two forms of the same apply
    asm volatile("" : "+v"(out));
    return out;
}

template <typename T> struct H16;
template <> struct H16<unsigned short> {
    __device__ static __forceinline__ unsigned int pack2(f32x2 v) { return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2)); }
    __device__ static __forceinline__ float lo(unsigned int w) { return __uint_as_float(w << 16); }
    __device__ static __forceinline__ float hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct H16<vs_half> {
    __device__ static __forceinline__ unsigned int pack2(f32x2 v) { return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2)); }
    __device__ static __forceinline__ float lo(unsigned int w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
    __device__ static __forceinline__ float hi(unsigned int w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
};

// four consecutive channels of one voxel: store rounded to T / widen what was loaded as dwords (4 for fp32, 2 for the 16-bit formats)
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        u32x2 pk;
        pk[0] = H16<T>::pack2(f32x2{v[0], v[1]});
        pk[1] = H16<T>::pack2(f32x2{v[2], v[3]});
        *(u32x2*)p = pk;
    }
}
template <typename T>
__device__ __forceinline__ void widen4(const unsigned int* w, float (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]); v[2] = __uint_as_float(w[2]); v[3] = __uint_as_float(w[3]);
    } else {
        v[0] = H16<T>::lo(w[0]); v[1] = H16<T>::hi(w[0]); v[2] = H16<T>::lo(w[1]); v[3] = H16<T>::hi(w[1]);
    }
}

// relu(x * scale + shift) on one 16-byte fragment of 8 channels (bf16 or fp16)
template <typename T>
__device__ __forceinline__ u32x4 act8(const u32x4 raw, const f32x2 (&sc)[4], const f32x2 (&sh)[4]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 v;
        v[0] = H16<T>::lo(raw[i]);
        v[1] = H16<T>::hi(raw[i]);
        v = v * sc[i] + sh[i];
        i16x2 s = __builtin_bit_cast(i16x2, H16<T>::pack2(v));
        s = __builtin_elementwise_max(s, i16x2{0, 0});  // ReLU on the bit patterns: in both formats negative values (and -0) are negative int16
        r[i] = __builtin_bit_cast(unsigned int, s);
    }
    return r;
}

// ---- fp32 parity mode on the bf16 matrix cores (igemm_k3x.h, wgrad.hip g3x_kernel) ----
// three-way split of four fp32 values into bf16 limbs, packed two per dword: out[l][0] = (v0, v1), out[l][1] = (v2, v3) of limb l.
// Round-to-nearest limbs (v_cvt_pk_bf16_f32), not truncation: x0 = rne(x), x1 = rne(x - x0), x2 = rne(x - x0 - x1); the subtractions are exact and
// x0 + x1 + x2 = x up to 2^-25 |x|.  With truncated limbs every limb has the sign of x, so the dropped products x1*w2 + x2*w1 + x2*w2 all have the
// sign of x*w: a 3e-8 relative BIAS that adds up coherently in the per-channel sums of a whole volume (seen: 8.6e-5 on the 96^3 statistics check);
// rounded limbs make the dropped terms zero-mean.
__device__ __forceinline__ void vs_limb_split4(const float (&v)[4], unsigned int (&out)[3][2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 x = f32x2{v[2 * q], v[2 * q + 1]};
        const unsigned int p0 = H16<unsigned short>::pack2(x);
        const f32x2 r1 = x - f32x2{H16<unsigned short>::lo(p0), H16<unsigned short>::hi(p0)};
        const unsigned int p1 = H16<unsigned short>::pack2(r1);
        const f32x2 r2 = r1 - f32x2{H16<unsigned short>::lo(p1), H16<unsigned short>::hi(p1)};
        out[0][q] = p0; out[1][q] = p1; out[2][q] = H16<unsigned short>::pack2(r2);
    }
}

// In-register limb arithmetic of the direct-from-global fp32 kernels (g1_kernel<float, ..., LIMB>; round 5 — in k3s_kernel<float>, where a B fragment is used
// once per wave and nothing amortises its split, the same arithmetic measured SLOWER: 6.205 -> 6.287 ms, profiles/r05_ab_fp32_k3s_limbs.json): the fp32 fragments of TWO
// consecutive k-groups (4 k-values per lane each — for A and B alike: any lane -> k assignment is valid as long as both operands share it) become three
// bf16 limb fragments of 8 k-values per lane, and one (A pair, B pair) product is six v_mfma_f32_16x16x32_bf16 (96 matrix cycles) instead of eight
// v_mfma_f32_16x16x4_f32 (256).  `small` takes the five products below the leading one (its own accumulator where the caller has the registers).
__device__ __forceinline__ void vs_limb_pair(const u32x4& f0, const u32x4& f1, u32x4 (&out)[3]) {
    float v0[4], v1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v0[j] = __uint_as_float(f0[j]); v1[j] = __uint_as_float(f1[j]); }
    unsigned int l0[3][2], l1[3][2];
    vs_limb_split4(v0, l0);
    vs_limb_split4(v1, l1);
#pragma unroll
    for (int l = 0; l < 3; ++l) out[l] = u32x4{l0[l][0], l0[l][1], l1[l][0], l1[l][1]};
}
__device__ __forceinline__ void vs_limb_mfma(const u32x4 (&al)[3], const u32x4 (&bl)[3], f32x4& lead, f32x4& small) {
    small = mfma16(al[0], bl[2], small, (unsigned short*)nullptr);
    small = mfma16(al[1], bl[1], small, (unsigned short*)nullptr);
    small = mfma16(al[2], bl[0], small, (unsigned short*)nullptr);
    small = mfma16(al[0], bl[1], small, (unsigned short*)nullptr);
    small = mfma16(al[1], bl[0], small, (unsigned short*)nullptr);
    lead = mfma16(al[0], bl[0], lead, (unsigned short*)nullptr);
}

// (bf16, 3x3x3, 8 stored k-side channels, <= 8 rows) weights — the 8-channel full-resolution layers — are packed in the Toeplitz
// fragment order of k3t_kernel (igemm_k3t.h): [k-group (tz,ty)][lane][8], row (lane & 15) = (dx2, co), k = (xpos = lane >> 4, ci),
// value W[co][ci][tz][ty][xpos - dx2] or 0.  pack.hip (image) and igemm_k3_bf16.hip (dispatch) both key on this predicate.
static __host__ __device__ inline bool vs_k3_toeplitz(int rows, int c_pad, int ntaps, int dtype) {
    return (dtype == VS_BF16 || dtype == VS_F16) && ntaps == 27 && c_pad == 8 && rows <= 8;
}
// The same layers in fp32 (k3_kernel<float, 8, 16, EPI, 4, true>, igemm_k3.h) use a Toeplitz layout along y: [k-group][lane][4],
// row (lane & 15) = (dy2, co), 36 window taps t = (dz, wy, dx) with wy = 0..3, k-group kg holds taps 2 kg and 2 kg + 1,
// k = ((lane >> 5) = which of the two, ci = 4 * ((lane >> 4) & 1) + j), value W[co][ci][dz][wy - dy2][dx] or 0: 18 k-groups.
static __host__ __device__ inline bool vs_k3_toeplitz_f32(int rows, int c_pad, int ntaps, int dtype) {
    return dtype == VS_F32 && ntaps == 27 && c_pad == 8 && rows <= 8;
}

// Channel-chunk width of the fp32 limb kernels (igemm_k3x.h) and of their VS_F32X3 weight images (pack.hip): min(c_pad, VS_K3X_CK), VS_K3X_CK = 8 or 16
// (env, read once; default below).  8: a stage's three limb planes + weight block take 64 KB of LDS — two workgroups per CU, whose staging and MFMA
// phases overlap; 16: half as many stages per tile, 115 KB — one workgroup per CU.
static inline int vs_k3x_ck(int c_pad) {
    // measured on the fp32 96^3 step: 7.34 ms (8) vs 7.55 ms (16); round 5: 6.299 vs 6.384, and 16 only for the layers with >= 64 / 128 channels: 6.213 / 6.207 vs 6.213
    // (profiles/r05_ab_fp32_ck16_from.json: nothing)
    const int ck = vs_cfg().k3x_ck;
    const int w = ck == 8 ? 8 : 16;
    return c_pad < w ? c_pad : w;
}

// (VS_F32X3, 3x3x3, 8 stored k-side channels, <= 8 rows) — the 8-channel full-resolution layers of the fp32 parity mode — are packed in the Toeplitz limb
// order of k3xt_kernel (igemm_k3x.h): [k-group (tz,ty)][limb][lane][8], row (lane & 15) = (dx2, co), k = (xpos = lane >> 4, ci), value
// W[co][ci][tz][ty][xpos - dx2] or 0.  pack.hip (image) and igemm_k3x.hip (dispatch) both key on this predicate (VS_K3X_TOEPLITZ=0: off).
static inline bool vs_k3x_toeplitz(int rows, int c_pad, int ntaps) {
    const int on = vs_cfg().k3x_toeplitz;
    return on && ntaps == 27 && c_pad == 8 && rows <= 8;
}

// Zeroing as a kernel, never hipMemsetAsync: inside a replayed HIP graph a memset node was observed to run out of order with the
// kernel nodes around it after a host-side D2H copy (second test-time-training case, nondeterministic bias gradients); kernel nodes
// are ordered.  One workgroup is enough for the few hundred bytes zeroed here.
__global__ static void vs_zero_kernel(unsigned int* p, long long words) {
    for (long long i = threadIdx.x; i < words; i += blockDim.x) p[i] = 0u;
}
static inline hipError_t vs_zero_async(void* p, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(vs_zero_kernel, dim3(1), dim3(256), 0, stream, (unsigned int*)p, (long long)(bytes / 4));
    return hipGetLastError();
}

#define VS_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

// run f with a null pointer of the storage type as tag: f((float*)0) / f((unsigned short*)0) [bf16 bits] / f((vs_half*)0) [fp16]
template <typename F>
static inline void dispatch_t(int dtype, F&& f) {
    if (dtype == VS_F32) f((float*)nullptr);
    else if (dtype == VS_BF16) f((unsigned short*)nullptr);
    else f((vs_half*)nullptr);
}
#define TAG_T(tag) typename std::remove_pointer<decltype(tag)>::type

static inline int vs_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
