// 3x3x3 (pad 1) convolution of the SMALL volumes (up to 6^3: the deep levels, C >= 32), bf16 / fp16: forward and backward-data.
//
// k3b_kernel's 4x4x16 tile is mostly padding there (6^3: 28 % of the tile's columns are voxels, 3^3: 14 %), every wave re-reads the
// whole weight block from LDS, and the measured bound of those launches is the LDS read bandwidth of the MFMA phase (1.25 KB of
// ds_read_b128 per MFMA; two chunks per stage changed nothing).  Here
//   * columns are FLATTENED voxels of one sample: a workgroup owns 64 consecutive voxels (4 column groups, all real except the tail);
//   * the four waves share those columns and SPLIT the 27 taps of a chunk (wave w: taps w, w+4, ...): a wave's weight fragments are
//     its own (global -> registers, no LDS) and it reads a quarter of the B fragments — 112 KB of LDS reads per stage instead of 540;
//     the partial accumulators meet in LDS after the last stage and every wave finishes one column group;
//   * the whole zero-padded sample chunk ((D+2)(H+2)(W+2) voxels x 32 channels: 32 KB at 6^3) is staged per chunk, whatever the tile.
// Staging (buffer loads two stages ahead, normalise-on-load, swizzled parts), statistics and fused IN-backward sums follow k3b_kernel; results are summed in a fixed order (bitwise reproducible).
#pragma once
#include <stdlib.h>
#include "igemm.h"

#ifndef K3_TICK                // phase stamps exist in the 16-bit translation units only (igemm_k3b.h)
#define K3_TICK(i)
#define K3_TICK_INIT
#define K3_TICK_FLUSH
#endif

#define K3S_LDS_RED 0          // float[4][16][2]
#define K3S_LDS_TILE 512
// then: padded sample chunk [TV][64 B] (at least 16 KB: the cross-wave partials alias it), scale / shift tables [C] each

// TVC: compile-time bound of the padded voxel count (128: up to 3x3x3, 512: up to 6x6x6) -> staging fragments per thread
// HS: the input is a lazy activation (normalise + ReLU while staging) — compile-time, like every other condition on the staging path: a
// run-time test between a load and its use makes the compiler drain vmcnt(0), i.e. wait for the prefetched stages as well
// T: unsigned short (bf16 bits), vs_half (fp16) or float; last template argument (kernel-name prefix unchanged).  A STAGE is 64 bytes per voxel
// in every type: a 32-channel chunk of the 16-bit types, HALF a chunk (16 channels, one of the two k-groups the packed image holds per tap) of
// fp32 — the exact-f32 MFMA runs at 1/16 of the 16-bit rate, so for fp32 the tap split over the waves and the unpadded columns matter even more
// (k3_kernel at 6^3 x 128: 81 us, at 3^3 x 256: 142 us).
template <bool SUMS, int TVC, bool HS, typename T = unsigned short>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k3s_kernel(const G1Params p) {
    K3_TICK_INIT
    constexpr int NIT = TVC * 4 / 256;                   // 16-byte fragments per thread per stage
    constexpr int NWI = 7;                               // weight fragments per thread per stage (27 * 64 / 256)
    constexpr int NKW = 7;                               // taps per wave per chunk (wave w: w, w + 4, ...)
    constexpr int ES = (int)sizeof(T), EPL = 16 / ES, CHS = 64 / ES;     // element size, elements per fragment, channels per stage
    constexpr int SPC = 32 / CHS;                        // stages per 32-channel chunk of the packed weight image (1, or 2 for fp32)
    constexpr int NCG = TVC == 128 ? 2 : 4;              // 16-column groups that can hold voxels: up to 3x3x3 = 27 voxels fill two (the other two were 4.5 of 13 us of MFMA phase on padding)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3S_LDS_RED);
    char* s_tile = smem + K3S_LDS_TILE;
    const int PX = p.W + 2, PY = p.H + 2, TV = (p.D + 2) * PY * PX, V = p.D * p.H * p.W;
    constexpr int tile_bytes = TVC * 64 > 16384 ? TVC * 64 : 16384;
    float* s_scale = (float*)(s_tile + tile_bytes);
    float* s_shift = s_scale + p.C;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int n = blockIdx.x / p.tiles_per_sample, ct = blockIdx.x - n * p.tiles_per_sample;
    const int rb0 = blockIdx.y;                          // 16-row block
    constexpr bool has_stats = HS;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * V * p.C * ES));
    const int nst = p.nch * SPC;                         // stages
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    // a / d for 0 <= a < 2048, 1 <= d <= 10 (everything here is that small): one multiply instead of a ~45-instruction integer
    // division — 24 of those made the prologue the longest phase of the launch
    const float inv_px = 1.0f / (float)PX, inv_py = 1.0f / (float)PY, inv_w = 1.0f / (float)p.W, inv_h = 1.0f / (float)p.H;
    auto sdiv = [](int a, float inv_d) { return (int)(((float)a + 0.5f) * inv_d); };
    // ---- staging geometry: fragment b = 16-byte part (tid & 3) of padded voxel (tid >> 2) + 64 b ---------------------------
    const int part = tid & 3;
    int goff[NIT];                                       // byte offset in x of this fragment for chunk 0, -1 = padding
    unsigned int swzbits = 0;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int pv = (tid >> 2) + 64 * b;
        const int t2 = sdiv(pv, inv_px), px = pv - t2 * PX, pz = sdiv(t2, inv_py), py = t2 - pz * PY;
        const bool ok = pv < TV && px >= 1 && px <= p.W && py >= 1 && py <= p.H && pz >= 1 && pz <= p.D;
        goff[b] = ok ? ((((n * p.D + pz - 1) * p.H + py - 1) * p.W + px - 1) * p.C + part * EPL) * ES : -1;
        swzbits |= (unsigned int)((px >> 2) & 1) << b;
    }
    // weights: with the taps split over the waves, tap (wave + 4 i)'s A fragment is used by this wave only — it goes from global
    // straight into the lane's registers (through LDS it cost 7 ds_write_b128 + 7 ds_read_b128 per thread and stage for no reuse)
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int kg = (tid >> 6) + 4 * i < 27 ? (tid >> 6) + 4 * i : 26;
        w_off[i] = rb0 * (p.nch * 27 * 64 * SPC) + kg * (64 * SPC) + (tid & 63);          // + (st / SPC) * 27 * 64 * SPC + (st % SPC) * 64
    }
    // two stages in flight (registers): with the MFMA phase this short, a stage requested only one stage ahead arrived late every time
    u32x4 xv0[NIT], wv0[NWI], xv1[NIT], wv1[NWI];
    auto load_stage = [&](int ch, u32x4 (&xv)[NIT], u32x4 (&wv)[NWI]) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + (ch / SPC) * (27 * 64 * SPC) + (ch % SPC) * 64];
#pragma unroll
        for (int b = 0; b < NIT; ++b)
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, goff[b] >= 0 ? goff[b] + ch * 64 : -1, 0, 0));
    };
    auto write_stage = [&](int ch, const u32x4 (&xv)[NIT], const u32x4 (&wv)[NWI]) {
        f32x2 sc[4], sh[4];                              // 16-bit: 8 channels; fp32: 4 (sc[0..1], sh[0..1])
        if (has_stats) {
#pragma unroll
            for (int i = 0; i < EPL / 2; ++i) {
                sc[i] = *(const f32x2*)(s_scale + ch * CHS + part * EPL + 2 * i);
                sh[i] = *(const f32x2*)(s_shift + ch * CHS + part * EPL + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            u32x4 v = xv[b];
            if (has_stats) {
                u32x4 a;
                if constexpr (ES == 2) a = act8<T>(v, sc, sh);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned int w = v[i];      // (a bit_cast applied to the vector element itself is mis-compiled: common.h, fp16 unpack)
                        const float scale = i & 1 ? sc[i >> 1][1] : sc[i >> 1][0], shift = i & 1 ? sh[i >> 1][1] : sh[i >> 1][0];
                        a[i] = __float_as_uint(fmaxf(__uint_as_float(w) * scale + shift, 0.f));
                    }
                }
                const bool ok = goff[b] >= 0;             // zero padding applies to the normalised activation
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            const int pv = (tid >> 2) + 64 * b;           // < TVC: the tile region holds TVC voxels, fragments beyond TV are zeros
            const int pw = part ^ (int)(((swzbits >> b) & 1u) << 1);
            *(u32x4*)(s_tile + pv * 64 + pw * 16) = v;
        }
    };

    // ---- first stage in flight; tables; per-lane read offsets ----------------------------------------------------------------
    load_stage(0, xv0, wv0);
    if (nst > 1) load_stage(1, xv1, wv1);
    if (has_stats) {
        for (int c = tid; c < p.C; c += 256) {
            float m, r;
            stats_to_mean_rstd_fast(p.x_stats, (size_t)n * p.C + c, (size_t)p.N * p.C, p.inv_count_in, p.eps, m, r);
            s_scale[c] = r; s_shift[c] = -m * r;
        }
    }
    const int row0 = rb0 * 16 + 4 * g;                   // first of this lane's 4 accumulator rows
    float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SUMS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) {
                stats_to_mean_rstd_fast(p.mask_stats, (size_t)n * p.M + row0 + r, (size_t)p.N * p.M, p.inv_count_out, p.eps, mm[r], mr[r]);
            }
    }
    if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = row0 + r < p.M ? p.bias[row0 + r] : 0.f;
    }
    // B fragment of (tap kg = wave + 4 i, column group cg): padded voxel (z + dz, y + dy, x + dx) of column voxel (z, y, x), part g
    // stored at part ^ ((px >> 2) & 1) << 1
    int boff[NKW][NCG];
    {
        int cz[NCG], cy[NCG], cx[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            int v = ct * 64 + cg * 16 + col;
            if (v >= V) v = 0;
            const int t2 = sdiv(v, inv_w);
            cx[cg] = v - t2 * p.W;
            cz[cg] = sdiv(t2, inv_h);
            cy[cg] = t2 - cz[cg] * p.H;
        }
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            int kg = wave + 4 * i;
            if (kg > 26) kg = 26;
            const int dz = kg / 9, dy = (kg / 3) % 3, dx = kg % 3;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                const int px = cx[cg] + dx;
                boff[i][cg] = (((cz[cg] + dz) * PY + cy[cg] + dy) * PX + px) * 64 + ((g ^ (((px >> 2) & 1) << 1)) * 16);
            }
        }
    }
    f32x4 acc[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                     // tables visible
    K3_TICK(0);

    u32x4 wa[NKW];                                       // the current stage's A fragments (the stage registers are re-requested before the MFMAs)
    auto multiply = [&]() {
        // no branches: a wave whose 7th tap does not exist (waves 3: taps 3, 7, ..., 27) multiplies a zeroed A fragment, so the
        // scheduler sees one block and keeps the next taps' LDS reads in flight under the MFMAs
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            u32x4 a = wa[i];
            if (i == NKW - 1) {
                const unsigned int keep = wave + 4 * i < 27 ? 0xffffffffu : 0u;
                a[0] &= keep; a[1] &= keep; a[2] &= keep; a[3] &= keep;
            }
            u32x4 b[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) b[cg] = *(const u32x4*)(s_tile + boff[i][cg]);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) acc[cg] = mfma16(a, b[cg], acc[cg], (T*)nullptr);
        }
    };
    for (int ch = 0; ch < nst; ch += 2) {
        if (ch > 0) __syncthreads();                     // every wave is done reading the previous stage
        K3_TICK(1);
        write_stage(ch, xv0, wv0);
        K3_TICK(2);
        __syncthreads();
        K3_TICK(3);
#pragma unroll
        for (int i = 0; i < NKW; ++i) wa[i] = wv0[i];
        if (ch + 2 < nst) load_stage(ch + 2, xv0, wv0);
        K3_TICK(4);
        multiply();
        K3_TICK(5);
        if (ch + 1 < nst) {
            __syncthreads();
            K3_TICK(1);
            write_stage(ch + 1, xv1, wv1);
            K3_TICK(2);
            __syncthreads();
            K3_TICK(3);
#pragma unroll
            for (int i = 0; i < NKW; ++i) wa[i] = wv1[i];
            if (ch + 3 < nst) load_stage(ch + 3, xv1, wv1);
            K3_TICK(4);
            multiply();
            K3_TICK(5);
        }
    }

    // ---- the four waves' partial sums meet in LDS (the tile is dead); wave w finishes column group w ---------------------------------
    __syncthreads();
    f32x4* s_part = (f32x4*)s_tile;                      // [wave][cg][lane]
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) s_part[(wave * NCG + cg) * 64 + lane] = acc[cg];
    __syncthreads();
    const int fcg = wave < NCG ? wave : 0;               // waves beyond the last column group finish nothing (valid = false below)
    f32x4 o = s_part[(0 * NCG + fcg) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 q = s_part[(w * NCG + fcg) * 64 + lane];
        o[0] += q[0]; o[1] += q[1]; o[2] += q[2]; o[3] += q[3];
    }

    // ---- epilogue: column voxel v of sample n, rows row0 .. row0 + 3 ---------------------------------------------------------------------
    const int v = ct * 64 + wave * 16 + col;
    const bool valid = v < V && row0 < p.M && wave < NCG;
    const int e = ((n * V + v) * p.M + row0) * ES;       // byte offset in y (and in the mask tensor)
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * V * p.M * ES));
    float xv4[4] = {0.f, 0.f, 0.f, 0.f};                 // SUMS: the mask tensor's values under this lane's outputs
    if constexpr (SUMS) {
        const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * V * p.M * ES));
        if constexpr (ES == 2) {
            const u32x2 mk = __builtin_bit_cast(u32x2, vs_raw_buffer_load_b64(mrsrc, valid ? e : -1, 0, 0));
            xv4[0] = H16<T>::lo(mk[0]); xv4[1] = H16<T>::hi(mk[0]);
            xv4[2] = H16<T>::lo(mk[1]); xv4[3] = H16<T>::hi(mk[1]);
        } else {
            const f32x4 mk = __builtin_bit_cast(f32x4, vs_raw_buffer_load_b128(mrsrc, valid ? e : -1, 0, 0));
            xv4[0] = mk[0]; xv4[1] = mk[1]; xv4[2] = mk[2]; xv4[3] = mk[3];
        }
    }
    float sv[4];                                         // 16-bit: round once to T; the statistics are those of the stored values
    if constexpr (ES == 2) {
        f32x2 lo, hi;
        lo[0] = o[0] + bv[0]; lo[1] = o[1] + bv[1];
        hi[0] = o[2] + bv[2]; hi[1] = o[3] + bv[3];
        i32x2 pk;
        pk[0] = (int)H16<T>::pack2(lo);
        pk[1] = (int)H16<T>::pack2(hi);
        vs_raw_buffer_store_b64(pk, yrsrc, valid ? e : -1, 0, 0);
        sv[0] = H16<T>::lo((unsigned int)pk[0]); sv[1] = H16<T>::hi((unsigned int)pk[0]);
        sv[2] = H16<T>::lo((unsigned int)pk[1]); sv[3] = H16<T>::hi((unsigned int)pk[1]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = o[r] + bv[r];
        vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{sv[0], sv[1], sv[2], sv[3]}), yrsrc, valid ? e : -1, 0, 0);
    }
    if (!valid) { sv[0] = 0.f; sv[1] = 0.f; sv[2] = 0.f; sv[3] = 0.f; }
    float ssum[4], ssq[4];
    if constexpr (SUMS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (xv4[r] - mm[r]) * mr[r];
            const float gm = xh > 0.f ? sv[r] : 0.f;
            ssum[r] = gm; ssq[r] = gm * xh;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[r] = sv[r]; ssq[r] = sv[r] * sv[r]; }
    }
    double* const red_dst0 = SUMS ? p.sums : p.y_stats;
    double* const red_dst = red_dst0;
    if (red_dst != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = ssum[r], q = ssq[r];
            { s = row16_sum(s); q = row16_sum(q); }
            if (col == 0) {
                s_red[(wave * 16 + 4 * g + r) * 2 + 0] = s;
                s_red[(wave * 16 + 4 * g + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int lr = tid >> 1, st = tid & 1, row = rb0 * 16 + lr;
            if (row < p.M) {
                const double tot = (double)s_red[(0 * 16 + lr) * 2 + st] + (double)s_red[(1 * 16 + lr) * 2 + st] +
                                   (double)s_red[(2 * 16 + lr) * 2 + st] + (double)s_red[(3 * 16 + lr) * 2 + st];
                stat_add(red_dst, (size_t)n * p.M + row, (size_t)p.N * p.M, st, tot);
            }
        }
    }
    K3_TICK(6);
    K3_TICK_FLUSH;
}

// volumes this kernel takes (C a multiple of 32):
static inline bool k3s_takes(const G1Params& p, int ck) {
    static const int on = getenv("VS_K3_SMALL") ? atoi(getenv("VS_K3_SMALL")) : 1;
    // up to 6x6x6 (padded sample chunk <= 512 voxels): at 8^3 (128^3 inputs) the 64 KB chunk and its 16 fragments per thread made this kernel
    // no faster than k3b_kernel (4.51 vs 4.47 ms per 128^3 domain-adaptation step)
    return on && ck == 32 && p.C % 32 == 0 && (p.D + 2) * (p.H + 2) * (p.W + 2) <= 512 && p.C <= 1024;
}

template <typename T, bool SUMS, int TVC, bool HS>
static int k3s_launch_t(const G1Params& p, int ctiles, hipStream_t stream) {
    const size_t lds = K3S_LDS_TILE + (size_t)(TVC * 64 > 16384 ? TVC * 64 : 16384) + (size_t)2 * p.C * sizeof(float);
    if (lds > 160 * 1024) return VS_ESHAPE;
    auto kern = k3s_kernel<SUMS, TVC, HS, T>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(ctiles * p.N, (p.M + 15) / 16), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T, bool SUMS>
static int k3s_launch(const G1Params& p_in, hipStream_t stream) {
    G1Params p = p_in;
    if (SUMS != (p.sums != nullptr) || (SUMS && p.x_stats != nullptr)) return VS_EINVAL;
    const int V = p.D * p.H * p.W, TV = (p.D + 2) * (p.H + 2) * (p.W + 2);
    const int ctiles = (V + 63) / 64;
    p.tiles_per_sample = ctiles;
    const bool hs = !SUMS && p.x_stats != nullptr;
#define K3S_GO(TVC) return hs ? k3s_launch_t<T, SUMS, TVC, !SUMS>(p, ctiles, stream) : k3s_launch_t<T, SUMS, TVC, false>(p, ctiles, stream)
    if (TV <= 128 && V <= 32) K3S_GO(128);
    if (TV <= 512) K3S_GO(512);
    return VS_ESHAPE;
#undef K3S_GO
}
