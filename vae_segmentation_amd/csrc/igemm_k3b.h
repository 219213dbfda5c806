// 3x3x3 (pad 1) implicit-GEMM convolution, bf16 operands / fp32 accumulate: forward and backward-data.
//
// Second-generation kernel for the bf16 path (the fp32 path keeps k3_kernel, igemm_k3.h).  Phase stamps of k3_kernel
// (tools/stamps_k3.py) showed that no phase was bound by MFMA or HBM; every phase paid a memory round trip instead:
//   * weight fragments came straight from global inside the MFMA loop, two loads in flight, queued (vmcnt retires in
//     issue order) behind the next tile's activation prefetch;
//   * bounds-tested staging compiled to exec-mask branches, whose joins made the compiler drain vmcnt(0) before every
//     epilogue store and inside the prefetch itself;
//   * normalise-on-load cost ~45 VALU per 16-byte fragment, repeated by every row-block workgroup over the 2.5x halo.
// Here a stage = (tile, channel chunk) is fetched whole into registers one stage ahead — activations through
// bounds-checked *buffer* loads (out-of-volume fragments read as zero, no branches), weights as the workgroup's
// [row block][k-group][lane] fragment block, loaded cooperatively once instead of once per wave — and both go through LDS.
// The MFMA loop then reads A and B from LDS with immediate offsets only (no VALU, no vmcnt), the halo tile of the 32-channel
// chunks is XOR-swizzled so every ds_read_b128 lane group covers all 16 slots of a bank row, and normalise+ReLU runs as
// packed fp32 fma / packed bf16 max (~24 VALU per fragment).  Epilogues, statistics and fragment formats are those of
// k3_kernel.
#pragma once
#include <stdlib.h>
#include "igemm.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// raw buffer load: lanes whose byte offset is >= num_records return 0 (hardware bounds check, stride 0)
__device__ i32x4 vs_raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned int bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned int)a;
    r[1] = (int)(unsigned int)((a >> 32) & 0xffffu);     // stride 0, no swizzle
    r[2] = (int)bytes;
    r[3] = 0x00020000;                                   // gfx9 raw buffer, 32-bit data format
    return r;
}

// relu(x * scale + shift) on one 16-byte fragment of 8 bf16 channels
__device__ __forceinline__ u32x4 act8(const u32x4 raw, const f32x2 (&sc)[4], const f32x2 (&sh)[4]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 v;
        v[0] = __uint_as_float(raw[i] << 16);
        v[1] = __uint_as_float(raw[i] & 0xffff0000u);
        v = v * sc[i] + sh[i];
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        i16x2 s = __builtin_bit_cast(i16x2, h);
        s = __builtin_elementwise_max(s, i16x2{0, 0});  // ReLU on the bf16 bit patterns: negative floats are negative int16
        r[i] = __builtin_bit_cast(unsigned int, s);
    }
    return r;
}

#define K3B_LDS_RED 0          // float[4][64][2]
#define K3B_LDS_TAPS 2048      // int[64]: byte offset of (k-group, lane group)'s tap in the halo tile (C = 8 / 16)
#define K3B_LDS_TILE 2304      // halo tile, weight block, then the per-(n,c) tables

template <int CK, int MT>
struct K3BGeom {
    static constexpr int RB = MT / 16;
    static constexpr bool SMALLC = CK < 32;                                  // several taps per 32-wide k-group, one chunk
    static constexpr int NKGC = SMALLC ? (27 * CK + 31) / 32 : 27;           // k-groups per channel chunk
    static constexpr int CKB = CK * 2;
    static constexpr int TILE_BYTES = 648 * CKB;
    static constexpr int NWF = RB * NKGC * 64;                               // 16-byte weight fragments per chunk per workgroup
    static constexpr int W_BYTES = NWF * 16;
};

// register budget: 4 workgroups per CU (one wave per SIMD each) for the small-channel kernels, 2 for the 32-channel chunks
// SUMS: backward-data use (input = a materialised gradient, no statistics; epilogue accumulates the fused IN-backward sums)
template <int CK, int MT, int EPI, bool SUMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CK == 32 ? (MT == 32 ? 1 : 2) : (MT == 32 ? (SUMS ? 2 : 3) : (CK == 16 && SUMS ? 3 : 4)), 8))) void k3b_kernel(const G1Params p) {
    typedef unsigned short T;
    using GEO = K3BGeom<CK, MT>;
    static_assert(CK == 8 || CK == 16 || CK == 32, "chunk width");
    constexpr int RB = GEO::RB, NKGC = GEO::NKGC, CKB = GEO::CKB, NWF = GEO::NWF;
    constexpr bool SMALLC = GEO::SMALLC;
    constexpr int U = CKB / 16;                          // 16-byte fragments per staged voxel
    constexpr int NU = 648 * U;
    constexpr int NIT = (NU + 255) / 256;                // activation fragments per thread per stage
    constexpr int NWI = (NWF + 255) / 256;               // weight fragments per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3B_LDS_RED);
    int* s_taps = (int*)(smem + K3B_LDS_TAPS);
    char* s_tile = smem + K3B_LDS_TILE;
    char* s_w = s_tile + GEO::TILE_BYTES;
    float* s_scale = (float*)(s_w + GEO::W_BYTES);       // rstd and -mean*rstd of the lazy input, [N*C] each
    float* s_shift = s_scale + p.N * p.C;
    float* s_mkm = s_shift + p.N * p.C;                  // mean / rstd of the mask tensor's channels (fused IN-bwd sums)
    float* s_mkr = s_mkm + p.N * p.M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    const bool has_stats = !SUMS && p.x_stats != nullptr;
    constexpr bool has_sums = SUMS;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 2));
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    // ---- per-thread stage geometry (tile independent) -----------------------------------------------------------------
    // fragment b of this thread is 16-byte part `part` (the same for every b: 256 % U == 0) of tile voxel tv_b
    const int part = tid % U;
    // LDS byte address of fragment b = lds_w0 + b * 4096 (tv advances by 256 / U voxels of CKB bytes per b), with the
    // 32-channel tile's swizzle bit of fragment b kept in swzbits
    int rel_off[NIT], tzyx[NIT];
    const int lds_w0 = (tid / U) * CKB;
    unsigned int swzbits = 0;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u / U;
        const int tx_ = tv % 18, ty_ = (tv / 18) % 6, tz_ = tv / 108;
        rel_off[b] = (((tz_ * p.H + ty_) * p.W + tx_) * p.C + part * 8) * 2;              // bytes from the tile's (0,0,0) halo voxel
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;                 // out-of-list fragments fail every bounds test
        if (CK == 32) swzbits |= (unsigned int)((tx_ >> 2) & 1) << b;                     // see baddr[] below
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int f = tid + i * 256;
        if (f > NWF - 1) f = NWF - 1;
        const int rb = f / (NKGC * 64), r = f - rb * (NKGC * 64);
        w_off[i] = (rb0 + rb) * (p.nch * NKGC * 64) + r;                                   // + ch * NKGC * 64
    }

    u32x4 xv[NIT], wv[NWI];
    unsigned int okbits = 0;
    auto tile_origin = [&](int t, int& n, int& z0, int& y0, int& x0) {
        n = t / p.tiles_per_sample;
        const int tl = t - n * p.tiles_per_sample;
        x0 = (tl % p.txn) * 16;
        y0 = ((tl / p.txn) % p.tyn) * 4;
        z0 = (tl / (p.txn * p.tyn)) * 4;
    };
    auto load_w = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + ch * (NKGC * 64)];
    };
    auto load_x = [&](int t, int ch) {
        int n, z0, y0, x0;
        tile_origin(t, n, z0, y0, x0);
        const int base = ((((n * p.D + z0 - 1) * p.H + y0 - 1) * p.W + x0 - 1) * p.C + ch * CK) * 2;
        okbits = 0;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int gz = z0 - 1 + (tzyx[b] & 0xff), gy = y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = x0 - 1 + (tzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            okbits |= ok ? (1u << b) : 0u;
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
        }
    };
    auto write_x = [&](int n, int ch) {
        f32x2 sc[4], sh[4];
        if (has_stats) {
            const int c0 = n * p.C + ch * CK + part * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_scale + c0 + 2 * i);
                sh[i] = *(const f32x2*)(s_shift + c0 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            u32x4 v = xv[b];
            if (has_stats) {
                const u32x4 a = act8(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;       // zero padding applies to the normalised activation
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            const int pw = CK == 32 ? (part ^ (int)(((swzbits >> b) & 1u) << 1)) : part;
            if (b < NIT - 1 || tid + b * 256 < NU) *(u32x4*)(s_tile + lds_w0 + b * 4096 + pw * 16) = v;
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            if (i < NWI - 1 || tid + i * 256 < NWF) *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    };

    // ---- first stage in flight before anything else; the statistics tables meanwhile ----------------------------------
    int t = blockIdx.x;                                  // the grid never exceeds the tile count
    load_w(0);
    load_x(t, 0);
    if (has_stats) {
        for (int i = tid; i < p.N * p.C; i += 256) {
            float m, r;
            stats_to_mean_rstd(p.x_stats + (size_t)i * 2, p.inv_count_in, p.eps, m, r);
            s_scale[i] = r;
            s_shift[i] = -m * r;
        }
    }
    if (has_sums) {
        for (int i = tid; i < p.N * p.M; i += 256) {
            float m, r;
            stats_to_mean_rstd(p.mask_stats + (size_t)i * 2, p.inv_count_out, p.eps, m, r);
            s_mkm[i] = m;
            s_mkr[i] = r;
        }
    }

    // ---- LDS read addresses ---------------------------------------------------------------------------------------------
    // B fragment of (tap, column voxel (wave, cg, col)): tile voxel (wave + dz, cg + dy, col + dx).
    //  CK == 32: a lane reads 16-byte part g of the voxel; parts are stored XOR ((tx >> 2) & 1) << 1, which puts the 16 lanes
    //            of every ds_read_b128 lane group on 16 different slots of the 256-byte bank row for dx = 0, 1, 2
    //            (unswizzled: 2-way conflicts on every read).  Per-lane base per dx, everything else an immediate.
    //  CK < 32 : the lane's tap depends on g; per-lane byte offset per k-group, cg is an immediate.
    int baddr[3];
    if constexpr (SMALLC) {
        if (tid < NKGC * 4) {
            int tap = (tid >> 2) * (32 / CK) + ((tid & 3) * 8) / CK;
            if (tap > 26) tap = 13;                      // padded taps read the centre voxel (their weights are zero)
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            s_taps[tid] = ((dz * 6 + dy) * 18 + dx) * CKB + (((tid & 3) * 8) % CK) * 2;
        }
        baddr[0] = (wave * 108 + col) * CKB;
    } else {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) baddr[dx] = (wave * 108 + col) * CKB + ((g ^ ((((col + dx) >> 2) & 1) << 1)) * 16);
    }
    const char* s_wl = s_w + lane * 16;

    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }

    constexpr bool restage_w = !SMALLC;                  // C = 8 / 16 is a single chunk: the weight block is staged once
    bool first = true;
    if constexpr (SMALLC) write_w();
    __syncthreads();                                     // tables visible

    for (; t < total_tiles; t += gridDim.x) {
        int n, z0, y0, x0;
        tile_origin(t, n, z0, y0, x0);
        const int oz = z0 + wave;
        f32x4 acc[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 mk[RB][4];                                 // mask tensor values under this tile's outputs (fused IN-bwd sums)

        for (int ch = 0; ch < p.nch; ++ch) {
            if (!first) __syncthreads();                 // every wave is done reading the previous stage
            write_x(n, ch);
            if constexpr (restage_w) write_w();
            first = false;
            __syncthreads();
            // requests of the next stage, oldest-needed first (vmcnt retires in issue order)
            const bool last_ch = ch + 1 == p.nch;
            if constexpr (EPI == EPI_RAW) {
                if (has_sums && last_ch) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const int row = (rb0 + rb) * 16 + 4 * g;
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) {
                            const int oy = y0 + cg, ox = x0 + col;
                            const bool valid = row < p.M && oz < p.D && oy < p.H && ox < p.W;
                            const size_t e = valid ? ((((size_t)n * p.D + oz) * p.H + oy) * p.W + ox) * p.M + row : 0;
                            mk[rb][cg] = *(const u32x2*)((const unsigned short*)p.mask_x + e);
                        }
                    }
                }
            }
            {
                const int tn = last_ch ? t + (int)gridDim.x : t;
                if (tn < total_tiles) {
                    if constexpr (restage_w) load_w(last_ch ? 0 : ch + 1);
                    load_x(tn, last_ch ? 0 : ch + 1);
                }
            }

            // ---- multiply this stage out of LDS ----
            // fragments of k-group kg+1 are read while k-group kg is multiplied; the scheduling barriers keep the compiler
            // from hoisting the whole unrolled loop's reads (hundreds of live registers) in front of the first MFMA
            auto read_kg = [&](int kg, u32x4 (&a)[RB], u32x4 (&b)[4]) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = *(const u32x4*)(s_wl + (rb * NKGC + kg) * 1024);
                if constexpr (SMALLC) {
                    const int o = baddr[0] + s_taps[kg * 4 + g];
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) b[cg] = *(const u32x4*)(s_tile + o + cg * 18 * CKB);
                } else {
                    const int dz = kg / 9, dy = (kg / 3) % 3, dx = kg % 3;
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) b[cg] = *(const u32x4*)(s_tile + baddr[dx] + (((dz * 6 + dy + cg) * 18 + dx) * CKB));
                }
            };
            u32x4 fa[2][RB], fb[2][4];
            read_kg(0, fa[0], fb[0]);
#pragma unroll
            for (int kg = 0; kg < NKGC; ++kg) {
                if (kg + 1 < NKGC) read_kg(kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = mfma16(fa[kg & 1][rb], fb[kg & 1][cg], acc[rb][cg], (T*)nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of this tile ----
        if constexpr (EPI == EPI_SOFTMAX2) {
            if (g == 0) {
                const float b0 = p.bias ? p.bias[0] : 0.f, b1 = p.bias ? p.bias[1] : 0.f;
                const size_t V = (size_t)p.D * p.H * p.W;
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const int oy = y0 + cg, ox = x0 + col;
                    if (!(oz < p.D && oy < p.H && ox < p.W)) continue;
                    float l0 = acc[0][cg][0] + b0, l1 = acc[0][cg][1] + b1;
                    const size_t v = ((size_t)oz * p.H + oy) * p.W + ox;
                    if (p.drop_p > 0.f) {
                        l0 *= dropout_scale(p.drop_seed, ((unsigned long long)n * 2 + 0) * V + v, p.drop_p);
                        l1 *= dropout_scale(p.drop_seed, ((unsigned long long)n * 2 + 1) * V + v, p.drop_p);
                    }
                    const float mx = fmaxf(l0, l1);
                    const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
                    const float inv = 1.f / (e0 + e1);
                    p.prob[((size_t)n * 2 + 0) * V + v] = e0 * inv;
                    p.prob[((size_t)n * 2 + 1) * V + v] = e1 * inv;
                }
            }
        } else {
            T* __restrict__ yout = (T*)p.y;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = (rb0 + rb) * 16 + 4 * g;
                const bool rvalid = row < p.M;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && rvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = p.bias[row + r];
                }
                float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f};
                if (has_sums && rvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * p.M + row + r]; mr[r] = s_mkr[n * p.M + row + r]; }
                }
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const int oy = y0 + cg, ox = x0 + col;
                    const bool valid = rvalid && oz < p.D && oy < p.H && ox < p.W;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = round_bf(acc[rb][cg][r] + bv[r]);
                    if (valid) {
                        const size_t e = ((((size_t)n * p.D + oz) * p.H + oy) * p.W + ox) * p.M + row;
                        u32x2 pk;
                        pk[0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
                        pk[1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
                        *(u32x2*)(yout + e) = pk;
                    }
                    const float keep = valid ? 1.f : 0.f;
                    if (has_sums) {
                        const u32x2 xx = mk[rb][cg];
                        float xv4[4];
                        xv4[0] = __uint_as_float(xx[0] << 16); xv4[1] = __uint_as_float(xx[0] & 0xffff0000u);
                        xv4[2] = __uint_as_float(xx[1] << 16); xv4[3] = __uint_as_float(xx[1] & 0xffff0000u);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (xv4[r] - mm[r]) * mr[r];
                            const float gm = xh > 0.f ? v[r] * keep : 0.f;
                            ssum[rb][r] += gm; ssq[rb][r] += gm * xh;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float vk = v[r] * keep; ssum[rb][r] += vk; ssq[rb][r] += vk * vk; }
                    }
                }
            }
            double* const red_dst = has_sums ? p.sums : p.y_stats;
            if (red_dst != nullptr) {
                const int tn = t + (int)gridDim.x;
                const bool flush = tn >= total_tiles || tn / p.tiles_per_sample != n;     // workgroup-uniform
                if (flush) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s = ssum[rb][r], q = ssq[rb][r];
#pragma unroll
                            for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                            if (col == 0) {
                                const int lr = rb * 16 + 4 * g + r;
                                s_red[(wave * 64 + lr) * 2 + 0] = s;
                                s_red[(wave * 64 + lr) * 2 + 1] = q;
                            }
                            ssum[rb][r] = 0.f; ssq[rb][r] = 0.f;
                        }
                    __syncthreads();
                    if (tid < MT * 2) {
                        const int lr = tid >> 1, st = tid & 1;
                        const int row = rb0 * 16 + lr;
                        if (row < p.M) {
                            const double tot = (double)s_red[(0 * 64 + lr) * 2 + st] + (double)s_red[(1 * 64 + lr) * 2 + st] +
                                               (double)s_red[(2 * 64 + lr) * 2 + st] + (double)s_red[(3 * 64 + lr) * 2 + st];
                            atomicAdd(red_dst + ((size_t)n * p.M + row) * 2 + st, tot);
                        }
                    }
                    __syncthreads();                     // s_red is reused by a later flush
                }
            }
        }
    }
}

template <int CK, int MT, int EPI, bool SUMS>
static int k3b_launch(const G1Params& p, int tiles_total, int row_tiles, hipStream_t stream) {
    using GEO = K3BGeom<CK, MT>;
    const size_t tables = (size_t)2 * p.N * p.C * sizeof(float) + (p.sums ? (size_t)2 * p.N * p.M * sizeof(float) : 0);
    const size_t lds = K3B_LDS_TILE + (size_t)GEO::TILE_BYTES + GEO::W_BYTES + tables;
    if (lds > 160 * 1024) return VS_ESHAPE;
    if ((long long)p.N * p.D * p.H * p.W * p.C * 2 >= 4294967296ll) return VS_ESHAPE;     // buffer offsets are 32-bit bytes
    if (SUMS != (p.sums != nullptr) || (SUMS && p.x_stats != nullptr)) return VS_EINVAL;
    auto kern = k3b_kernel<CK, MT, EPI, SUMS>;
    // idempotent one-time opt-in to the full 160 KiB of dynamic LDS (not a stream operation)
    static const hipError_t attr_err =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    // persistent grid: a few workgroups per CU, each walking a strided slice of the tile list
    static const int per_cu = getenv("VS_K3_WGS_PER_CU") ? atoi(getenv("VS_K3_WGS_PER_CU")) : 4;   // tuning knob
    int wg = 256 * per_cu / (row_tiles < per_cu ? row_tiles : per_cu);
    if (wg < 256) wg = 256;
    const int gx = tiles_total < wg ? tiles_total : wg;
    hipLaunchKernelGGL(kern, dim3(gx, row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
