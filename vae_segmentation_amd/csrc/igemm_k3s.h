// 3x3x3 (pad 1) convolution of the SMALL volumes (up to 6^3: the deep levels, C >= 32), bf16 / fp16: forward and backward-data.
//
// k3b_kernel's 4x4x16 tile is mostly padding there (6^3: 28 % of the tile's columns are voxels, 3^3: 14 %), every wave re-reads the
// whole weight block from LDS, and the measured bound of those launches is the LDS read bandwidth of the MFMA phase (1.25 KB of
// ds_read_b128 per MFMA; two chunks per stage changed nothing).  Here
//   * columns are FLATTENED voxels of one sample: a workgroup owns 64 consecutive voxels (4 column groups, all real except the tail);
//   * the four waves share those columns and SPLIT the 27 taps of a chunk (wave w: taps w, w+4, ...): a wave's weight fragments are
//     its own (global -> registers, no LDS) and it reads a quarter of the B fragments — 112 KB of LDS reads per stage instead of 540;
//     the partial accumulators meet in LDS after the last stage and every wave finishes one column group;
//   * the whole zero-padded sample chunk ((D+2)(H+2)(W+2) voxels x 32 channels: 32 KB at 6^3) is in LDS per chunk, whatever the tile — its zero padding written
//     once per body, its REAL voxels loaded / normalised / written per stage (end of round 6), into two alternating buffers (one barrier per stage).
// Staging (buffer loads two stages ahead, normalise-on-load, swizzled parts), statistics and fused IN-backward sums follow k3b_kernel; results are summed in a fixed order (bitwise reproducible).
#pragma once
#include <stdlib.h>
#include "igemm.h"
#include "chain.h"

#ifndef K3_TICK                // phase stamps exist in the 16-bit translation units only (igemm_k3b.h)
#define K3_TICK(i)
#define K3_TICK_INIT
#define K3_TICK_FLUSH
#endif

// diagnostic build only (-DVS_CHAIN_STAMPS, tools/chain_stamps.py): 100 MHz time stamps of the first XCD slot's workgroups, 16 per layer
#if defined(VS_CHAIN_STAMPS) && defined(VS_CHAIN_STAMPS_TU)
__device__ unsigned long long g_chain_stamps[64 * 64];
extern "C" int vs_debug_read_chain_stamps(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chain_stamps), sizeof(unsigned long long) * n); }
#define CH_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64 && (i) < 64) g_chain_stamps[(blockIdx.x >> 3) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-phase shader-clock sums of the stage loop (phases 1..5 of KS_TICK) -> stamps sb + 10 .. 14, the body's total -> sb + 15
#define KS_TICK_INIT unsigned long long tk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk_last = __builtin_amdgcn_s_memtime(); const unsigned long long tk_first = tk_last;
#define KS_TICK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk_acc[i] += now_ - tk_last; tk_last = now_; } while (0)
#define KS_TICK_FLUSH do { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64) { for (int i_ = 1; i_ <= 5; ++i_) g_chain_stamps[(blockIdx.x >> 3) * 64 + sb + 9 + i_] = tk_acc[i_]; g_chain_stamps[(blockIdx.x >> 3) * 64 + sb + 15] = tk_last - tk_first; } } while (0)
#else
#define CH_STAMP(i)
#define KS_TICK_INIT K3_TICK_INIT
#define KS_TICK(i) K3_TICK(i)
#define KS_TICK_FLUSH K3_TICK_FLUSH
#endif

// K3S_NCG (chain.h): 16-voxel column groups per workgroup at the 6^3-class volumes.  2 since round 6 (32-voxel column tiles, twice the workgroups): these launches
// use 64 of the 256 CUs with 64-voxel tiles and their MFMA phase is issue-bound per CU — same-box 2.349 -> 2.331 ms per bf16 step, 6.096 -> 5.971 in the fp32
// mode (whose exact-f32 MFMA phase is the longest); 16-voxel tiles (K3S_NCG=1) lose again (2.401 vs 2.349: every workgroup stages the whole padded sample).
static inline int k3s_col_tile(bool small) { return small ? 16 * K3S_NCG_SMALL : 16 * K3S_NCG; }      // host: voxels per column tile
#define K3S_LDS_RED 0          // float[4][16][2]
#define K3S_LDS_TILE 512
// then: two padded sample chunks [2][TVC][64 B] (k3s_tile_bytes; at least 16 KB: the cross-wave partials alias them), scale / shift tables [C] each

// TVC: compile-time bound of the padded voxel count (128: up to 3x3x3, 512: up to 6x6x6) -> staging fragments per thread
// HS: the input is a lazy activation (normalise + ReLU while staging) — compile-time, like every other condition on the staging path: a
// run-time test between a load and its use makes the compiler drain vmcnt(0), i.e. wait for the prefetched stages as well
// T: unsigned short (bf16 bits), vs_half (fp16) or float; last template argument (kernel-name prefix unchanged).  A STAGE is 64 bytes per voxel
// in every type: a 32-channel chunk of the 16-bit types, HALF a chunk (16 channels, one of the two k-groups the packed image holds per tap) of
// fp32 — the exact-f32 MFMA runs at 1/16 of the 16-bit rate, so for fp32 the tap split over the waves and the unpadded columns matter even more
// (k3_kernel at 6^3 x 128: 81 us, at 3^3 x 256: 142 us).
// CH (chain.h): the body runs as one layer of a chain kernel — its input (and, forward, the input's statistics) may have been written by other
// workgroups of THIS launch: the weights of the first two stages are requested, then the workgroup waits for `wait_target` arrivals on `wait_ctr`
// (none when wait_target == 0), then everything handed over is loaded with sc1 loads; the output is stored write-through (sc1).
// XR: x stages in flight (registers).  The standalone kernels keep 2 (with the weights: a deeper ring measured slower there, round 3); the chain
// kernels request up to 4 (6^3) / 8 (3^3) stages of the hand-over tensor at once — behind a layer boundary its lines come from the memory side,
// not from L2, and a stage requested two stages ahead arrived late every time (tools/chain_stamps.py); the weights keep their ring of 2.
// NW: waves per workgroup (4, or 8 in the chain kernels).  These bodies are instruction-bound at one wave per SIMD with their phases in series
// (tools/chain_stamps.py: per stage 0.45 us of normalise + LDS write of the WHOLE padded sample, 0.45 us of LDS reads + MFMA, 0.35 us of barrier skew):
// with eight waves the staging pass and the 27 taps are split eight ways — two waves per SIMD interleave one another's phases.
// What a body's threads need to know about the volume and their column tile, and nothing about the layer: computed once per kernel (the chain kernels' layers share
// it; recomputed per layer it was ~0.6 us of every layer's prologue — tools/chain_stamps.py, "entry -> weights requested").
template <int TVC, int NW> struct K3SGeo {
    static constexpr int NT = 64 * NW;
    static constexpr int VRMAX = TVC == 128 ? 32 : 216;                  // real voxels: V <= 32 in the small class (k3s_launch), <= 6^3 where the padded volume is <= 512
    static constexpr int NIT = (VRMAX * 4 + NT - 1) / NT;                // 16-byte fragments per thread per stage
    static constexpr int NKW = (27 + NW - 1) / NW;                       // taps per wave per chunk
    static constexpr int NCG = TVC == 128 ? K3S_NCG_SMALL : K3S_NCG;    // 16-column groups per workgroup
    int loff[NIT];                                       // staging: LDS byte offset of fragment b in the padded tile, part swizzle included (no such voxel: padding voxel 0)
    int boff[NKW][NCG];                                  // B fragment of (tap wave + NW i, column group cg): LDS byte offset
};
template <int TVC, int NW>
__device__ __forceinline__ void k3s_geometry(K3SGeo<TVC, NW>& geo, const int D, const int H, const int W, const int ct) {
    using G = K3SGeo<TVC, NW>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4, part = tid & 3;
    const int PX = W + 2, PY = H + 2, V = D * H * W;
    // a / d for 0 <= a < 2048, 1 <= d <= 10 (everything here is that small): one multiply instead of a ~45-instruction integer
    // division — 24 of those made the prologue the longest phase of the launch
    const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
    auto sdiv = [](int a, float inv_d) { return (int)(((float)a + 0.5f) * inv_d); };
    // staging: fragment b = 16-byte part (tid & 3) of REAL voxel (tid >> 2) + (NT / 4) b of the sample
#pragma unroll
    for (int b = 0; b < G::NIT; ++b) {
        const int rv = (tid >> 2) + (G::NT / 4) * b;
        const int t2 = sdiv(rv, inv_w), vx = rv - t2 * W, vz = sdiv(t2, inv_h), vy = t2 - vz * H;
        const int px = vx + 1, pv = ((vz + 1) * PY + vy + 1) * PX + px;
        geo.loff[b] = rv < V ? pv * 64 + ((part ^ (((px >> 2) & 1) << 1)) * 16) : part * 16;
    }
    // B fragment of (tap kg = wave + NW i, column group cg): padded voxel (z + dz, y + dy, x + dx) of column voxel (z, y, x), part g
    // stored at part ^ ((px >> 2) & 1) << 1
    constexpr int CW = 16 * G::NCG;
    int cz[G::NCG], cy[G::NCG], cx[G::NCG];
#pragma unroll
    for (int cg = 0; cg < G::NCG; ++cg) {
        int v = ct * CW + cg * 16 + col;
        if (v >= V) v = 0;
        const int t2 = sdiv(v, inv_w);
        cx[cg] = v - t2 * W;
        cz[cg] = sdiv(t2, inv_h);
        cy[cg] = t2 - cz[cg] * H;
    }
#pragma unroll
    for (int i = 0; i < G::NKW; ++i) {
        int kg = wave + NW * i;
        if (kg > 26) kg = 26;
        const int dz = kg / 9, dy = (kg / 3) % 3, dx = kg % 3;
#pragma unroll
        for (int cg = 0; cg < G::NCG; ++cg) {
            const int px = cx[cg] + dx;
            geo.boff[i][cg] = (((cz[cg] + dz) * PY + cy[cg] + dy) * PX + px) * 64 + ((g ^ (((px >> 2) & 1) << 1)) * 16);
        }
    }
}

template <int TVC> static constexpr int k3s_tile_bytes() { return 2 * TVC * 64 > 16384 ? 2 * TVC * 64 : 16384; }      // two stage buffers; at least 16 KB (the cross-wave partials alias them)

template <bool SUMS, int TVC, bool HS, typename T, bool CH, int XR = 2, int NW = 4>
__device__ __forceinline__ void k3s_body(const G1Params& p, const K3SGeo<TVC, NW>& geo, const int n, const int ct, const int rb0, char* smem, unsigned int* wait_ctr = nullptr,
                                         unsigned int wait_target = 0, unsigned int* fault = nullptr, const int sb = 0 /* CH_STAMP base */) {
    KS_TICK_INIT
    CH_STAMP(sb + 0);
    constexpr int XAUX = CH ? VS_AUX_SC1 : 0;
    constexpr int NT = 64 * NW;                          // threads
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    // Only the REAL voxels of the padded sample are staged (end of round 6): the zero padding — 296 of the 512 voxels of a padded 6^3 sample, 98 of 125 at 3^3 — is
    // written to LDS ONCE per body instead of being loaded (offset -1), normalised and masked again in every stage by every workgroup of the sample.
    constexpr int NIT = K3SGeo<TVC, NW>::NIT;            // 16-byte fragments per thread per stage
    constexpr int NKW = (27 + NW - 1) / NW;              // taps per wave per chunk (wave w: w, w + NW, ...): 7 or 4
    constexpr int NWI = NKW;                             // weight fragments per thread per stage: its wave's taps
    constexpr int ES = (int)sizeof(T), EPL = 16 / ES, CHS = 64 / ES;     // element size, elements per fragment, channels per stage
    constexpr int SPC = 32 / CHS;                        // stages per 32-channel chunk of the packed weight image (1, or 2 for fp32)
    constexpr int NCG = TVC == 128 ? K3S_NCG_SMALL : K3S_NCG;        // 16-column groups that can hold voxels
    constexpr int CW = 16 * NCG;                         // voxels per column tile: up to 3x3x3 = 27 voxels fill two (the other two were 4.5 of 13 us of MFMA phase on padding)
    float* s_red = (float*)(smem + K3S_LDS_RED);
    char* s_tile = smem + K3S_LDS_TILE;
    const int PX = p.W + 2, PY = p.H + 2, TV = (p.D + 2) * PY * PX, V = p.D * p.H * p.W;
    // TWO tile buffers (end of round 6): stage ch goes to buffer ch & 1, so the write of stage ch + 1 need not wait until every wave has read stage ch — one barrier
    // per stage instead of two (with eight waves a barrier is ~0.17 us of arrival skew: tools/chain_stamps.py).  Safe with one: a wave writes stage ch + 1 into the
    // buffer stage ch - 1 was read from only after passing the barrier of stage ch, which every wave reaches after its MFMAs of stage ch - 1.
    constexpr int tile_bytes = k3s_tile_bytes<TVC>();
    float* s_scale = (float*)(s_tile + tile_bytes);
    float* s_shift = s_scale + p.C;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    constexpr bool has_stats = HS;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * V * p.C * ES));
    const int nst = p.nch * SPC;                         // stages
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    // ---- staging: fragment b = 16-byte part (tid & 3) of REAL voxel (tid >> 2) + (NT / 4) b of the sample (LDS offsets: geo.loff) ---------------------------
    const int part = tid & 3;
    int goff[NIT];                                       // byte offset in x of this fragment for chunk 0, -1 = no such voxel
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int rv = (tid >> 2) + (NT / 4) * b;
        goff[b] = rv < V ? ((n * V + rv) * p.C + part * EPL) * ES : -1;
    }
    static_assert((TVC * 4) % NT == 0, "zero fill: whole rounds of 16-byte stores");
    {
        // the padding (and everything the previous user of this LDS left there: cross-wave partials, apply tables): zeros, once per body — real voxels are rewritten by every stage
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 2 * TVC * 4 / NT; ++k) *(u32x4*)(s_tile + (tid + NT * k) * 16) = z4;      // both stage buffers
    }
    // weights: with the taps split over the waves, tap (wave + 4 i)'s A fragment is used by this wave only — it goes from global
    // straight into the lane's registers (through LDS it cost 7 ds_write_b128 + 7 ds_read_b128 per thread and stage for no reuse)
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int kg = (tid >> 6) + NW * i < 27 ? (tid >> 6) + NW * i : 26;
        w_off[i] = rb0 * (p.nch * 27 * 64 * SPC) + kg * (64 * SPC) + (tid & 63);          // + (st / SPC) * 27 * 64 * SPC + (st % SPC) * 64
    }
    // two stages in flight (registers): with the MFMA phase this short, a stage requested only one stage ahead arrived late every time
    static_assert(XR >= 2 && XR % 2 == 0, "x ring depth");
    u32x4 xv[XR][NIT], wv[2][NWI];
    auto load_w = [&](int ch, u32x4 (&wv)[NWI]) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + (ch / SPC) * (27 * 64 * SPC) + (ch % SPC) * 64];
    };
    auto load_x = [&](int ch, u32x4 (&xv)[NIT]) {
#pragma unroll
        for (int b = 0; b < NIT; ++b)
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, goff[b] >= 0 ? goff[b] + ch * 64 : -1, 0, XAUX));
    };
    auto write_stage = [&](int ch, const u32x4 (&xv)[NIT], const int buf) {
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
            *(u32x4*)(s_tile + buf * (TVC * 64) + geo.loff[b]) = v;
        }
    };

    // ---- first stage in flight; tables; per-lane read offsets ----------------------------------------------------------------
    double st_pre[2] = {0.0, 1.0};                       // chain: this thread's first (sum, sumsq) pair, requested ahead of the fragments (vmcnt retires in order)
    if constexpr (CH) {
        load_w(0, wv[0]);
        if (nst > 1) load_w(1, wv[1]);
        CH_STAMP(sb + 1);
        if (wait_target != 0) chain_wait(wait_ctr, wait_target, fault);
        CH_STAMP(sb + 2);
        if (has_stats && tid < p.C) stat_load_x<CH>(p.x_stats, (size_t)n * p.C + tid, (size_t)p.N * p.C, st_pre);
#pragma unroll
        for (int r = 0; r < XR; ++r)
            if (r < nst) load_x(r, xv[r]);
    } else {
        load_w(0, wv[0]); load_x(0, xv[0]);
        if (nst > 1) { load_w(1, wv[1]); load_x(1, xv[1]); }
    }
    if (has_stats) {
        for (int c = tid; c < p.C; c += NT) {
            float m, r;
            double sv[2] = {st_pre[0], st_pre[1]};
            if (!CH || c != tid) stat_load_x<CH>(p.x_stats, (size_t)n * p.C + c, (size_t)p.N * p.C, sv);
            stats_to_mean_rstd_fast(sv, p.inv_count_in, p.eps, m, r);
            s_scale[c] = r; s_shift[c] = -m * r;
        }
    }
    const int row0 = rb0 * 16 + 4 * g;                   // first of this lane's 4 accumulator rows
    float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    // SUMS: mean / rstd of the mask tensor's 16 channels under this workgroup's rows (forward statistics: an earlier launch's, cold lines).  They are needed in the
    // EPILOGUE only: thread t < 16 requests row t's four copies here — behind the x stages in the load queue, so no stage waits for them — and turns them into the
    // table entry after the stage loop.  (Every lane used to run the whole statistics -> mean / rstd chain for its four rows right here: 1.7-2 us of every backward
    // body's prologue went into waiting for these lines — tools/chain_stamps.py, "tables built".)
    double mraw[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    if constexpr (SUMS) {
        if (tid < 16 && rb0 * 16 + tid < p.M) stat_load_raw(p.mask_stats, (size_t)n * p.M + rb0 * 16 + tid, (size_t)p.N * p.M, mraw);
    }
    if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = row0 + r < p.M ? p.bias[row0 + r] : 0.f;
    }
    f32x4 acc[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                     // tables visible
    KS_TICK(0);
    if constexpr (CH) CH_STAMP(sb + 3);

    u32x4 wa[NKW];                                       // the current stage's A fragments (the stage registers are re-requested before the MFMAs)
    auto multiply = [&](const int buf) {
        // no branches: a wave whose 7th tap does not exist (waves 3: taps 3, 7, ..., 27) multiplies a zeroed A fragment, so the
        // scheduler sees one block and keeps the next taps' LDS reads in flight under the MFMAs
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            u32x4 a = wa[i];
            if (i == NKW - 1) {
                const unsigned int keep = wave + NW * i < 27 ? 0xffffffffu : 0u;
                a[0] &= keep; a[1] &= keep; a[2] &= keep; a[3] &= keep;
            }
            u32x4 b[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) b[cg] = *(const u32x4*)(s_tile + buf * (TVC * 64) + geo.boff[i][cg]);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) acc[cg] = mfma16(a, b[cg], acc[cg], (T*)nullptr);
        }
    };
    for (int ch0 = 0; ch0 < nst; ch0 += XR) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int ch = ch0 + r;
            if (ch >= nst) break;                         // workgroup-uniform
            KS_TICK(1);
            write_stage(ch, xv[r], r & 1);               // XR is even: ch & 1 == r & 1
            KS_TICK(2);
            __syncthreads();
            KS_TICK(3);
#pragma unroll
            for (int i = 0; i < NKW; ++i) wa[i] = wv[r & 1][i];
            if (ch + 2 < nst) load_w(ch + 2, wv[r & 1]);
            if (ch + XR < nst) load_x(ch + XR, xv[r]);
            KS_TICK(4);
            multiply(r & 1);
            KS_TICK(5);
        }
    }

    // ---- the four waves' partial sums meet in LDS (the tile is dead); wave w finishes column group w ---------------------------------
    if constexpr (CH) CH_STAMP(sb + 4);
    if constexpr (SUMS) {                                // the mask tensor's table (s_scale / s_shift are free: a backward body's input is a stored gradient)
        if (tid < 16) {
            float m = 0.f, r = 0.f;
            if (rb0 * 16 + tid < p.M) {
                double sv2[2];
                stat_combine(mraw, sv2);
                stats_to_mean_rstd_fast(sv2, p.inv_count_out, p.eps, m, r);
            }
            s_scale[tid] = m; s_shift[tid] = r;
        }
    }
    __syncthreads();
    f32x4* s_part = (f32x4*)s_tile;                      // [wave][cg][lane]
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) s_part[(wave * NCG + cg) * 64 + lane] = acc[cg];
    __syncthreads();
    if constexpr (SUMS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { mm[r] = s_scale[4 * g + r]; mr[r] = s_shift[4 * g + r]; }
    }
    const int fcg = wave < NCG ? wave : 0;               // waves beyond the last column group finish nothing (valid = false below)
    f32x4 o = s_part[(0 * NCG + fcg) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const f32x4 q = s_part[(w * NCG + fcg) * 64 + lane];
        o[0] += q[0]; o[1] += q[1]; o[2] += q[2]; o[3] += q[3];
    }

    // ---- epilogue: column voxel v of sample n, rows row0 .. row0 + 3 ---------------------------------------------------------------------
    const int v = ct * CW + wave * 16 + col;
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
        vs_raw_buffer_store_b64(pk, yrsrc, valid ? e : -1, 0, XAUX);
        sv[0] = H16<T>::lo((unsigned int)pk[0]); sv[1] = H16<T>::hi((unsigned int)pk[0]);
        sv[2] = H16<T>::lo((unsigned int)pk[1]); sv[3] = H16<T>::hi((unsigned int)pk[1]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = o[r] + bv[r];
        vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{sv[0], sv[1], sv[2], sv[3]}), yrsrc, valid ? e : -1, 0, XAUX);
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
            if (col == 0 && wave < 4) {                   // (waves 4 .. 7 of an eight-wave workgroup finish no column group: their sums are zero)
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
    KS_TICK(6);
    KS_TICK_FLUSH;
    if constexpr (CH) CH_STAMP(sb + 5);
}

// waves per workgroup: eight (round 6: 2.422 -> 2.387 ms per 96^3 step in the chain kernels of the 6^3 class, same box).  The 3^3 class ran four — its eight-wave build
// spilled — until the bodies stopped holding the padding's fragments in registers (end of round 6: 164-178 registers, no spills): eight there too, 2.2274 -> 2.2225 ms
// same box, the fp32 mode unchanged (profiles/r06_ab_k3s_eight_waves_3cube*.json).  Chain and standalone kernels use the same count: their results are bit-identical.
template <int TVC> struct K3SWaves { static constexpr int NW = 8; };

template <bool SUMS, int TVC, bool HS, typename T = unsigned short>
__global__ __launch_bounds__(64 * K3SWaves<TVC>::NW) __attribute__((amdgpu_waves_per_eu(K3SWaves<TVC>::NW / 4, 2))) void k3s_kernel(const G1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = blockIdx.x / p.tiles_per_sample, ct = blockIdx.x - n * p.tiles_per_sample;
    K3SGeo<TVC, K3SWaves<TVC>::NW> geo;
    k3s_geometry(geo, p.D, p.H, p.W, ct);
    k3s_body<SUMS, TVC, HS, T, false, 2, K3SWaves<TVC>::NW>(p, geo, n, ct, (int)blockIdx.y /* 16-row block */, smem);
}

// ---- chain kernels (chain.h): the convolutions of a DoubleConv at one of these volumes in ONE launch --------------------------------------------
// forward: layers 0 .. nl-1, each a 3x3x3 conv on the previous one's lazy output (layer 0: the chain's input, lazy or stored)
// backward (BWD): layers in backward order, each a backward-data conv with the fused IN-backward sums of ITS output's activation (p.sums != nullptr;
// then bit l of apply_mask: the apply pass runs in place before the next layer reads it) or a plain one (the stored input of the block)
template <int TVC, typename T, bool BWD, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, 2))) void k3s_chain_kernel(const K3Chain c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XRC = TVC == 128 ? 8 : 4;               // x stages in flight: 2 fragments per thread and stage at 3^3, 8 at 6^3
    const ChainPlace pl = chain_place(c);
    const int item = pl.item;
    if (item >= c.items) return;
    const int ctiles = c.p[0].tiles_per_sample, ct = item % ctiles, rb0 = item / ctiles;
    const unsigned int items = (unsigned int)c.items;
    K3SGeo<TVC, NW> geo;                                 // the layers of a chain share the volume and the column tile
    k3s_geometry(geo, c.p[0].D, c.p[0].H, c.p[0].W, ct);
    // Warm this XCD's L2 with every layer's weight rows of this workgroup (one dword per 128-byte line; the values are only consumed after the last
    // layer): a stage's weight request otherwise goes to memory — ~3 us per pair of stages in flight (tools/chain_stamps.py), the largest share of a layer.
    unsigned int warm[VS_CHAIN_MAX_LAYERS][32 / NW];
#pragma unroll
    for (int l = 0; l < VS_CHAIN_MAX_LAYERS; ++l) {
        const G1Params& p = c.p[l < c.nl ? l : 0];
        constexpr int SPCW = 32 / (64 / (int)sizeof(T));
        const int lines = p.nch * 27 * 8 * SPCW;          // 128-byte lines of one 16-row block of the packed image
        const char* base = (const char*)p.wp + (size_t)(rb0 < p.rb_total ? rb0 : 0) * lines * 128;
#pragma unroll
        for (int k = 0; k < 32 / NW; ++k) {
            const int ln = (int)threadIdx.x + 64 * NW * k;
            warm[l][k] = (l < c.nl && ln < lines) ? *(const unsigned int*)(base + (size_t)ln * 128) : 0u;
        }
    }
    for (int n = pl.n0; n < c.p[0].N; n += pl.nstep) {
        int phase = 0;
        for (int l = 0; l < c.nl; ++l) {
            const G1Params& p = c.p[l];
            unsigned int* wc = phase > 0 ? chain_counter(c, n, phase - 1) : nullptr;
            const unsigned int wt = phase > 0 ? items : 0u;
            const bool active = rb0 < p.rb_total;          // layers with fewer output rows than the widest one leave the last workgroups idle (they still arrive)
            if constexpr (BWD) {
                if (!active) {}
                else if (p.sums != nullptr) k3s_body<true, TVC, false, T, true, XRC, NW>(p, geo, n, ct, rb0, smem, wc, wt, c.fault, l * 16);
                else k3s_body<false, TVC, false, T, true, XRC, NW>(p, geo, n, ct, rb0, smem, wc, wt, c.fault, l * 16);
                if ((c.apply_mask >> l) & 1) {
                    chain_arrive(chain_counter(c, n, phase));
                    CH_STAMP(l * 16 + 6);
                    chain_wait(chain_counter(c, n, phase), items, c.fault);
                    CH_STAMP(l * 16 + 7);
                    ++phase;
                    chain_apply<T>(p.y, p.mask_x, p.mask_stats, p.sums, l + 1 == c.nl ? c.add : nullptr, n, p.N, p.D * p.H * p.W, p.M, p.inv_count_out, p.eps,
                                   item, c.items, (float*)(smem + K3S_LDS_TILE));
                    CH_STAMP(l * 16 + 8);
                }
            } else {
                if (!active) {}
                else if (p.x_stats != nullptr) k3s_body<false, TVC, true, T, true, XRC, NW>(p, geo, n, ct, rb0, smem, wc, wt, c.fault, l * 16);
                else k3s_body<false, TVC, false, T, true, XRC, NW>(p, geo, n, ct, rb0, smem, wc, wt, c.fault, l * 16);
            }
            if (l + 1 < c.nl) { chain_arrive(chain_counter(c, n, phase)); ++phase; }
            else __syncthreads();                        // the next sample of this slot reuses the LDS
            CH_STAMP(l * 16 + 9);
        }
    }
    unsigned int wx = 0;
#pragma unroll
    for (int l = 0; l < VS_CHAIN_MAX_LAYERS; ++l)
#pragma unroll
        for (int k = 0; k < 32 / NW; ++k) wx |= warm[l][k] & 0x7f800000u;
    if (wx == 0xffffffffu) atomicOr(c.fault, 2u);        // never true: keeps the warming loads alive
}

// volumes this kernel takes (C a multiple of 32):
static inline bool k3s_takes(const G1Params& p, int ck) {
    const int on = vs_cfg().k3_small;
    // up to 6x6x6 (padded sample chunk <= 512 voxels): at 8^3 (128^3 inputs) the 64 KB chunk and its 16 fragments per thread made this kernel
    // no faster than k3b_kernel (4.51 vs 4.47 ms per 128^3 domain-adaptation step)
    return on && ck == 32 && p.C % 32 == 0 && (p.D + 2) * (p.H + 2) * (p.W + 2) <= 512 && p.C <= 1024;
}

template <typename T, bool SUMS, int TVC, bool HS>
static int k3s_launch_t(const G1Params& p, int ctiles, hipStream_t stream) {
    const size_t lds = K3S_LDS_TILE + (size_t)k3s_tile_bytes<TVC>() + (size_t)2 * p.C * sizeof(float);
    if (lds > 160 * 1024) return VS_ESHAPE;
    auto kern = k3s_kernel<SUMS, TVC, HS, T>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(ctiles * p.N, (p.M + 15) / 16), dim3(64 * K3SWaves<TVC>::NW), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T, bool SUMS>
static int k3s_launch(const G1Params& p_in, hipStream_t stream) {
    G1Params p = p_in;
    if (SUMS != (p.sums != nullptr) || (SUMS && p.x_stats != nullptr)) return VS_EINVAL;
    const int V = p.D * p.H * p.W, TV = (p.D + 2) * (p.H + 2) * (p.W + 2);
    const bool small_v = TV <= 128 && V <= 32;
    const int ctiles = (V + k3s_col_tile(small_v) - 1) / k3s_col_tile(small_v);
    p.tiles_per_sample = ctiles;
    const bool hs = !SUMS && p.x_stats != nullptr;
#define K3S_GO(TVC) return hs ? k3s_launch_t<T, SUMS, TVC, !SUMS>(p, ctiles, stream) : k3s_launch_t<T, SUMS, TVC, false>(p, ctiles, stream)
    if (TV <= 128 && V <= 32) K3S_GO(128);
    if (TV <= 512) K3S_GO(512);
    return VS_ESHAPE;
#undef K3S_GO
}

// host side of k3s_chain_kernel: c.p[l] hold pointers, N, D, H, W, C, M, eps, inv_counts; geometry is filled here.  VS_ESHAPE when the chain does not fit this kernel.
static inline bool k3s_chain_takes(int d, int h, int w, int c_in) {
    return c_in % 32 == 0 && c_in <= 1024 && (long long)(d + 2) * (h + 2) * (w + 2) <= 512;
}
template <typename T>
static int k3s_chain_launch(K3Chain c, bool bwd, hipStream_t stream) {
    const G1Params& p0 = c.p[0];
    const int V = p0.D * p0.H * p0.W, TV = (p0.D + 2) * (p0.H + 2) * (p0.W + 2);
    const bool small_v = TV <= 128 && V <= 32;
    const int ctiles = (V + k3s_col_tile(small_v) - 1) / k3s_col_tile(small_v);
    int rbmax = 0, cmax = 0;
    for (int l = 0; l < c.nl; ++l) {
        G1Params& p = c.p[l];
        if (!k3s_chain_takes(p.D, p.H, p.W, p.C) || p.M % 8 || p.D != p0.D || p.H != p0.H || p.W != p0.W || p.N != p0.N) return VS_ESHAPE;
        if (bwd ? (p.x_stats != nullptr || (p.sums == nullptr && ((c.apply_mask >> l) & 1))) : (p.sums != nullptr)) return VS_EINVAL;
        p.tiles_per_sample = ctiles;
        p.nch = p.C / 32;
        p.rb_total = (p.M + 15) / 16;
        rbmax = p.rb_total > rbmax ? p.rb_total : rbmax;
        cmax = p.C > cmax ? p.C : cmax;
        cmax = p.M > cmax ? p.M : cmax;
    }
    c.items = ctiles * rbmax;
    const int grid = chain_grid(c);
    if (grid == 0) return VS_ESHAPE;
    const bool small = TV <= 128 && V <= 32;
    const int tvc = small ? 128 : 512;
    const size_t lds = K3S_LDS_TILE + (size_t)(small ? k3s_tile_bytes<128>() : k3s_tile_bytes<512>()) + (size_t)2 * cmax * sizeof(float);
    if (lds > 160 * 1024 || (size_t)4 * cmax * sizeof(float) > 16384) return VS_ESHAPE;
    int threads = 256;
    auto go = [&](auto kern) -> int {
        const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_err != hipSuccess) return (int)attr_err;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, c);
        VS_CHECK_LAUNCH();
        return VS_OK;
    };
    threads = small ? 64 * K3SWaves<128>::NW : 64 * K3SWaves<512>::NW;
    if (small) return bwd ? go(k3s_chain_kernel<128, T, true, K3SWaves<128>::NW>) : go(k3s_chain_kernel<128, T, false, K3SWaves<128>::NW>);
    return bwd ? go(k3s_chain_kernel<512, T, true, K3SWaves<512>::NW>) : go(k3s_chain_kernel<512, T, false, K3SWaves<512>::NW>);
}
