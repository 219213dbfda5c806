// Weight-gradient kernels for every conv kind:
//   dW[m][c][tap] = sum_{n,v} actP(P)[n,v,m] * actQ(Q)[n, v*s + off(tap) - p, c]
// GEMM orientation: D[row = m][col = (tap, c)] += sum_{k = voxel} A[m][k] * B[k][col], on the exact-f32
// MFMA (v_mfma_f32_16x16x4_f32) for both storage types: with one element per lane per operand the
// channels-last tiles are consumed as they lie in LDS (no transposition), and fp32 products keep the
// gradient exact for bf16 activations as well.
// Each wave accumulates over its share of voxels in registers and writes one partial slab; a second
// kernel sums the slabs in a fixed order (bitwise reproducible) straight into the reference's
// [m][c][tap] fp32 layout.
#include <stdlib.h>
#include "common.h"

enum { G3_K3 = 0, G3_K2S2 = 1, G3_UP = 2 };   // G3_UP: the composed Up block (igemm_k4.h): Q as K3 on the coarse grid, P = the FINE output gradient read space-to-depth
#define G3_MAXN 16

struct G3Params {
    const void* P; const double* P_stats;
    const void* Q; const double* Q_stats;
    float* ws;
    int N, Dp, Hp, Wp;        // P grid (the voxel loop runs over it)
    int Dq, Hq, Wq;           // Q grid
    int Mch, Cch;             // stored channels of P / Q
    int mbn, cbn;             // 16-row blocks of m, CB-channel blocks of c
    int ksplit, total_tiles, tiles_per_sample, tyn, txn;
    float eps;
    double inv_cnt_p, inv_cnt_q;
    int up_co;                // G3_UP: channels of the fine tensor P points to (Mch = 8 * up_co is its space-to-depth view)
    unsigned int fd_m[3];     // multiply-shift division (common.h fdiv) by tiles_per_sample, txn * tyn, txn: the tile coordinates of every loop
    unsigned int fd_s;        //   round cost ~12 scalar instructions instead of ~100 (five 32-bit divisions); shifts packed 8 bits each
    int variant;              // which instantiation of g3b_body a workgroup of the all-buckets launch runs (g3b_uber_kernel): G3V_*
    int mp;                   // 8-channel 3x3x3 layers with 8 stored P channels, 16-bit grouped path: MFMA rows 8..15 carry P shifted by one voxel in x
                              //   (see g3b_body) — 9 column blocks (dz, dy) x {dx = 0, +1} instead of 14 blocks of two taps
};

// (m, s) with n / d == (mulhi(n, m) + n) >> s for every 0 <= n < 2^31 (as igemm_k3b.h k3b_fastdiv)
static inline void g3_fastdiv(G3Params& p) {
    const int d[3] = {p.tiles_per_sample, p.txn * p.tyn, p.txn};
    p.fd_s = 0;
    for (int i = 0; i < 3; ++i) {
        unsigned int sh = 0;
        while ((1ll << sh) < d[i]) ++sh;
        p.fd_m[i] = (unsigned int)((((1ull << (32 + sh)) + (unsigned long long)d[i] - 1) / (unsigned long long)d[i]) - (1ull << 32));
        p.fd_s |= sh << (8 * i);
    }
}
// tile t -> sample n, tile origin (z0, y0, x0)
__device__ __forceinline__ void g3_tile_coords(const G3Params& p, int t, int& n, int& z0, int& y0, int& x0) {
    n = fdiv(t, p.fd_m[0], p.fd_s & 0xff);
    const int tl = t - n * p.tiles_per_sample;
    const int tz = fdiv(tl, p.fd_m[1], (p.fd_s >> 8) & 0xff);
    const int r = tl - tz * p.txn * p.tyn;
    const int ty = fdiv(r, p.fd_m[2], (p.fd_s >> 16) & 0xff);
    const int tx = r - ty * p.txn;
    z0 = tz * 4; y0 = ty * 4; x0 = tx * 16;
}

template <int CB, int KIND> struct G3Geo {
    static constexpr int NTAPS = KIND != G3_K2S2 ? 27 : 8;
    static constexpr int NCB = CB == 16 ? NTAPS : (NTAPS + 1) / 2;
    static constexpr int QZ = KIND != G3_K2S2 ? 6 : 8, QY = KIND != G3_K2S2 ? 6 : 8, QX = KIND != G3_K2S2 ? 18 : 32;
    static constexpr int QV = QZ * QY * QX;
};

#define G3_LDS_STATS 0                 // 4 x float[G3_MAXN][16]
#define G3_LDS_P (4 * G3_MAXN * 16 * 4)
#define G3_LDS_Q (G3_LDS_P + 256 * 16 * 4)


// Sum the four waves' accumulators in a fixed order (w0+w1+w2+w3: bitwise reproducible) through LDS and write ONE
// partial slab per workgroup: ws[ks][mb*cbn+cb][k][col][row].  `s_buf` needs NCB*256 floats.
template <int NCB>
__device__ __forceinline__ void g3_finish(f32x4 (&acc)[NCB], float* s_buf, float* ws_tile, int wave, int col, int g, int nk = NCB) {
    float* mine = s_buf + col * 16 + 4 * g;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                if (k < nk) {                             // uniform: the M-packed form of g3b_body uses the first 9 blocks
                    f32x4 v = acc[k];
                    if (w > 0) {
                        const f32x4 o = *(const f32x4*)(mine + k * 256);
                        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                    }
                    if (w < 3) *(f32x4*)(mine + k * 256) = v;
                    else *(f32x4*)(ws_tile + (size_t)k * 256 + col * 16 + 4 * g) = v;
                }
            }
        }
    }
}

template <typename T, int CB, int KIND>
__device__ __forceinline__ void g3_body(const G3Params& p, const int bx, const int ks) {
    using GEO = G3Geo<CB, KIND>;
    constexpr int NTAPS = GEO::NTAPS, NCB = GEO::NCB, QY = GEO::QY, QX = GEO::QX, QV = GEO::QV;
    constexpr int EPL = ET<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_pm = (float*)(smem + G3_LDS_STATS);
    float* s_pr = s_pm + G3_MAXN * 16;
    float* s_qm = s_pr + G3_MAXN * 16;
    float* s_qr = s_qm + G3_MAXN * 16;
    float* s_p = (float*)(smem + G3_LDS_P);
    float* s_q = (float*)(smem + G3_LDS_Q);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int mb = bx / p.cbn, cb = bx - mb * p.cbn;
    const bool p_stats = p.P_stats != nullptr, q_stats = p.Q_stats != nullptr;
    const i32x4 prsrc = make_rsrc(p.P, (unsigned int)((long long)p.N * p.Dp * p.Hp * p.Wp * p.Mch * (int)sizeof(T)));
    const i32x4 qrsrc = make_rsrc(p.Q, (unsigned int)((long long)p.N * p.Dq * p.Hq * p.Wq * p.Cch * (int)sizeof(T)));

    // ---- tile-independent fragment geometry (as g3b_body): the next tile's P and Q fragments are requested — bounds-checked buffer loads, no
    // branches — right after the current tile went to LDS, so their latency is covered by the MFMA phase (the round-1 form loaded, staged and
    // multiplied each tile in turn, with ~80 integer operations of address arithmetic per fragment: 180 us per 96^3 layer for 80 us of MFMA cycles)
    constexpr int PU = 16 / EPL, QU = CB / EPL;           // 16-byte fragments per P / Q voxel
    constexpr int NP = PU;                                // P fragments per thread (256 voxels x PU / 256 threads)
    constexpr int NQ = QV * QU, NITQ = (NQ + 255) / 256;
    const int ppart = tid % PU, qpart = tid % QU;         // constant per thread (256 % PU == 256 % QU == 0)
    const bool pch_ok = mb * 16 + ppart * EPL < p.Mch, qch_ok = cb * CB + qpart * EPL < p.Cch;
    int prel[NP], pzyx[NP];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        const int v = (tid + b * 256) / PU;
        const int lx = v & 15, ly = (v >> 4) & 3, lz = v >> 6;
        prel[b] = (((lz * p.Hp + ly) * p.Wp + lx) * p.Mch + mb * 16 + ppart * EPL) * (int)sizeof(T);
        pzyx[b] = pch_ok ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    int qrel[NITQ], qzyx[NITQ];
#pragma unroll
    for (int b = 0; b < NITQ; ++b) {
        const int u = tid + b * 256;
        const int v = u / QU;
        const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
        qrel[b] = (((lz * p.Hq + ly) * p.Wq + lx) * p.Cch + cb * CB + qpart * EPL) * (int)sizeof(T);
        qzyx[b] = (u < NQ && qch_ok) ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    u32x4 pv[NP], qv[NITQ];
    unsigned long long okbits = 0;                        // bit b: P fragment b inside the volume; bit NP + b: Q fragment b (up to 4 + 32 of them)
    auto request = [&](int t) {
        int n, z0, y0, x0;
        g3_tile_coords(p, t, n, z0, y0, x0);
        okbits = 0;
        const int pbase = (((n * p.Dp + z0) * p.Hp + y0) * p.Wp + x0) * p.Mch * (int)sizeof(T);
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const int gz = z0 + (pzyx[b] & 0xff), gy = y0 + ((pzyx[b] >> 8) & 0xff), gx = x0 + (pzyx[b] >> 16);
            const bool ok = gz < p.Dp && gy < p.Hp && gx < p.Wp;
            okbits |= ok ? (1ull << b) : 0ull;
            pv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(prsrc, ok ? pbase + prel[b] : -1, 0, 0));
        }
        const int qz0 = KIND == G3_K3 ? z0 - 1 : 2 * z0, qy0 = KIND == G3_K3 ? y0 - 1 : 2 * y0, qx0 = KIND == G3_K3 ? x0 - 1 : 2 * x0;
        const int qbase = (((n * p.Dq + qz0) * p.Hq + qy0) * p.Wq + qx0) * p.Cch * (int)sizeof(T);
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const int gz = qz0 + (qzyx[b] & 0xff), gy = qy0 + ((qzyx[b] >> 8) & 0xff), gx = qx0 + (qzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.Dq && (unsigned)gy < (unsigned)p.Hq && (unsigned)gx < (unsigned)p.Wq;
            okbits |= ok ? (1ull << (NP + b)) : 0ull;
            qv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(qrsrc, ok ? qbase + qrel[b] : -1, 0, 0));
        }
    };
    auto commit = [&](int n) {                            // registers -> (normalised) fp32 LDS tiles; out-of-volume fragments are zeros
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            float f[EPL];
            frag_unpack(pv[b], f, (T*)nullptr);
            const bool ok = (okbits >> b) & 1ull;
#pragma unroll
            for (int jj = 0; jj < EPL; ++jj) {
                if (p_stats) {
                    const float tt = (f[jj] - s_pm[n * 16 + ppart * EPL + jj]) * s_pr[n * 16 + ppart * EPL + jj];
                    f[jj] = tt > 0.f ? tt : 0.f;
                }
                f[jj] = ok ? f[jj] : 0.f;
            }
            const int v = (tid + b * 256) / PU;
#pragma unroll
            for (int jj = 0; jj < EPL; jj += 4)
                *(f32x4*)(s_p + v * 16 + ppart * EPL + jj) = f32x4{f[jj], f[jj + 1], f[jj + 2], f[jj + 3]};
        }
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            float f[EPL];
            frag_unpack(qv[b], f, (T*)nullptr);
            const bool ok = (okbits >> (NP + b)) & 1ull;
#pragma unroll
            for (int jj = 0; jj < EPL; ++jj) {
                if (q_stats) {
                    const float tt = (f[jj] - s_qm[n * 16 + qpart * EPL + jj]) * s_qr[n * 16 + qpart * EPL + jj];
                    f[jj] = tt > 0.f ? tt : 0.f;
                }
                f[jj] = ok ? f[jj] : 0.f;
            }
            const int u = tid + b * 256;
            if (b < NITQ - 1 || u < NQ) {
#pragma unroll
                for (int jj = 0; jj < EPL; jj += 4)
                    *(f32x4*)(s_q + (u / QU) * CB + qpart * EPL + jj) = f32x4{f[jj], f[jj + 1], f[jj + 2], f[jj + 3]};
            }
        }
    };

    int t = ks;                                           // ksplit never exceeds the tile count
    request(t);
    // mean / rstd tables for this WG's channel blocks, all samples
    for (int i = tid; i < p.N * 16; i += 256) {
        const int n = i >> 4, c = i & 15;
        float m = 0.f, r = 1.f;
        const int pc = mb * 16 + c;
        if (p_stats && pc < p.Mch) stats_to_mean_rstd(p.P_stats, (size_t)n * p.Mch + pc, (size_t)p.N * p.Mch, p.inv_cnt_p, p.eps, m, r);
        s_pm[i] = m; s_pr[i] = r;
        m = 0.f; r = 1.f;
        const int qc = cb * CB + c;
        if (q_stats && c < CB && qc < p.Cch) stats_to_mean_rstd(p.Q_stats, (size_t)n * p.Cch + qc, (size_t)p.N * p.Cch, p.inv_cnt_q, p.eps, m, r);
        s_qm[i] = m; s_qr[i] = r;
    }

    // per-lane B-side tap offsets (in Q-tile voxels) for each column block
    int qoff[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        int tap = CB == 16 ? k : 2 * k + (col >> 3);
        if (tap >= NTAPS) tap = KIND == G3_K3 ? 13 : 0;     // padded column: reads a valid voxel, result discarded
        int dz, dy, dx;
        if (KIND == G3_K3) { dz = tap / 9; dy = (tap / 3) % 3; dx = tap % 3; }
        else { dz = (tap >> 2) & 1; dy = (tap >> 1) & 1; dx = tap & 1; }
        qoff[k] = (dz * QY + dy) * QX + dx;
    }
    const int cq = CB == 16 ? col : (col & 7);

    f32x4 acc[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (; t < p.total_tiles; t += p.ksplit) {
        __syncthreads();      // previous tile fully consumed (also orders the stats tables on the first pass)
        commit(fdiv(t, p.fd_m[0], p.fd_s & 0xff));
        __syncthreads();
        if (t + p.ksplit < p.total_tiles) request(t + p.ksplit);
        // ---- 64 voxels of this wave's z-slice, 4 per MFMA step ----
#pragma unroll 2
        for (int step = 0; step < 16; ++step) {
            const int ly = step >> 2, lx = (step & 3) * 4 + g;
            const float a = s_p[((wave * 4 + ly) * 16 + lx) * 16 + col];
            int qbase;
            if (KIND == G3_K3) qbase = ((wave * QY + ly) * QX + lx);
            else qbase = ((2 * wave * QY + 2 * ly) * QX + 2 * lx);
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                const float b = s_q[(qbase + qoff[k]) * CB + cq];
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
            }
        }
    }

    const size_t slab_elems = (size_t)p.mbn * p.cbn * NCB * 256;
    g3_finish<NCB>(acc, s_p, p.ws + (size_t)ks * slab_elems + ((size_t)bx * NCB) * 256, wave, col, g);
}

template <typename T, int CB, int KIND>
__global__ __launch_bounds__(256) void g3_kernel(const G3Params p) { g3_body<T, CB, KIND>(p, blockIdx.x, blockIdx.y); }

// ---------------------------------------------------------------------------------------------------
// bf16 storage: the same GEMM on v_mfma_f32_16x16x32_bf16.  The MFMA wants 8 consecutive k (= voxels) per lane
// for a fixed row/column (= channel) while the tiles are channels-last, so the operands are read with
// ds_read_b64_tr_b16: per 16-lane group it fetches a 4-voxel x 16-channel block and hands lane i channel i of
// the 4 voxels.  k-mapping of a K-step (32 voxels = two x-rows of the wave's 4x16 z-slice): hardware
// k = 8g + 4r + e  <->  voxel (y = 2s + r, x = 4g + e); A and B use the same map, so the sum is unchanged,
// and the four lane groups of one read touch 512 contiguous bytes (conflict-free).
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ u32x4 tr_pair(const char* s_base, int off0, int off1) {       // eight 16-bit k-values of one row / column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_base + off1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(u32x4, v);
}

#define G3B_LDS_P (4 * G3_MAXN * 16 * 4)
#define G3B_LDS_Q (G3B_LDS_P + 256 * 16 * 2)

// Tile t+ksplit's P and Q fragments are requested (bounds-checked buffer loads: out-of-volume fragments read as zero,
// no branches) right after tile t went to LDS, so their latency is covered by tile t's MFMA phase; the lazy operands'
// normalise+ReLU runs as packed fma / packed max (common.h act8).
// T: unsigned short (bf16 bits) or vs_half (fp16) — the transposing LDS read moves 16-bit words, whatever they encode
template <typename T, int CB, int KIND, bool MP = false>
__device__ __forceinline__ void g3b_body(const G3Params& p, const int bx, const int ks) {
    using GEO = G3Geo<CB, KIND>;
    constexpr int NTAPS = GEO::NTAPS, NCB = GEO::NCB, QY = GEO::QY, QX = GEO::QX, QV = GEO::QV;
    constexpr int QROW = CB * 2;                 // bytes per Q-tile voxel
    constexpr int QU = CB / 8;                   // 16-byte fragments per Q-tile voxel
    constexpr int NQ = QV * QU;
    constexpr int NITQ = (NQ + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_psc = (float*)(smem + G3_LDS_STATS);     // scale (rstd) / shift (-mean * rstd) of P's and Q's 16 channels, per sample
    float* s_psh = s_psc + G3_MAXN * 16;
    float* s_qsc = s_psh + G3_MAXN * 16;
    float* s_qsh = s_qsc + G3_MAXN * 16;
    char* s_p = smem + G3B_LDS_P;                // [256 voxels][16 ch] bf16
    char* s_q = smem + G3B_LDS_Q;                // [QV voxels][CB ch] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int q4 = col >> 2, p4 = col & 3;       // tr-read addressing: this lane supplies row q4, columns 4*p4..4*p4+3
    const int mb = bx / p.cbn, cb = bx - mb * p.cbn;
    const bool p_stats_in = p.P_stats != nullptr, q_stats_in = p.Q_stats != nullptr;
#ifdef VS_G3B_ABLATE                                 // diagnostic build (tools/build_stamps.sh): parts of the tile loop switched off to see what the loop time is made of
    const int ablate = KIND == G3_K3 ? p.up_co : 0;      // 1: no MFMA phase, 4: no normalise, 8: no loads after the first tile, 16: no LDS writes, 32: no barriers
#else
    constexpr int ablate = 0;
#endif
    const i32x4 prsrc = make_rsrc(p.P, (unsigned int)((long long)p.N * p.Dp * p.Hp * p.Wp * p.Mch * 2));
    const i32x4 qrsrc = make_rsrc(p.Q, (unsigned int)((long long)p.N * p.Dq * p.Hq * p.Wq * p.Cch * 2));

    // ---- tile-independent fragment geometry ----
    // P fragment b: voxel v = (tid + 256 b) >> 1 of the 4x4x16 tile, channel half `ppart`; Q fragment b: voxel qv of the halo /
    // strided region, 16-byte part `qpart` (constant per thread: 256 % QU == 0)
    const int ppart = tid & 1, qpart = tid % QU;
    // M-packed form (p.mp: 8 stored P channels, 8-channel Q, 3x3x3): the upper half of the 16 MFMA rows — padding otherwise — carries the SAME 8 channels
    // one voxel further in x, rows (s, co): sum_v P(v + s) Q(v + t) = dW[t - s].  With column taps dx in {0, +1} the rows s = 0 give dx = 0, +1 and the rows
    // s = 1 give dx = -1 (and dx = 0 again, dropped by the reduction): 9 column blocks (dz, dy) x {dx = 0 | +1} x 8 channels instead of 14 blocks of two
    // taps — 18 instead of 28 MFMAs and 36 instead of 56 transposing LDS reads per tile and wave (the kernel is bound by LDS bandwidth and instruction issue).
    // The shifted fragment is loaded straight from global memory into the slot of channels 8..15, so the P tile needs no halo.
    // 16-channel Q blocks (a 16 -> 8 layer): the same rows against the taps dx in {0, +1} of all 16 channels — 18 blocks (dz, dy, dx select) instead of 27.
    // (a kernel instantiation of its own, MP: with both forms behind a run-time branch the 16-channel kernel needed 408 registers instead of 236)
    constexpr bool mp = MP && KIND == G3_K3;
    const int pshift = mp ? ppart : 0, pchan = mp ? 0 : ppart * 8;
    const bool pch_ok = mp || mb * 16 + ppart * 8 < p.Mch, qch_ok = cb * CB + qpart * 8 < p.Cch;
    // bounds tests in limit form: fragment b lies inside the volume iff z0 < plz[b] && y0 < ply[b] && x0 < plx[b] (tile origin scalar, limits per lane:
    // three compares per fragment instead of unpacking packed local coordinates every tile — the kernel is bound by instruction issue)
    int prel[2], plz[2], ply[2], plx[2];
    // G3_UP: view channel block mb of coarse voxel v = 16 channels of the fine tensor [N][2Dp][2Hp][2Wp][Co] around voxel 2v:
    // Co = 8: mb = (pz, py), the block's halves are the x parities; Co = 16: mb = parity; Co >= 32: mb = (parity, 16-channel block)
    const int up_fh = 2 * p.Hp, up_fw = 2 * p.Wp;
    int up_mb_off = 0;
    if constexpr (KIND == G3_UP) {
        const int co = p.up_co;
        if (co == 8) up_mb_off = ((((mb >> 1) & 1) * up_fh + (mb & 1)) * up_fw) * co * 2;
        else {
            const int nb16 = co / 16, pp = mb / nb16, cb16 = mb - pp * nb16;
            up_mb_off = ((((pp >> 2) & 1) * up_fh + ((pp >> 1) & 1)) * up_fw + (pp & 1)) * co * 2 + cb16 * 32;
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int v = (tid + b * 256) >> 1;
        const int lx = v & 15, ly = (v >> 4) & 3, lz = v >> 6;
        if constexpr (KIND == G3_UP) prel[b] = (((2 * lz) * up_fh + 2 * ly) * up_fw + 2 * lx) * p.up_co * 2 + up_mb_off + ppart * 16;
        else prel[b] = (((lz * p.Hp + ly) * p.Wp + lx + pshift) * p.Mch + mb * 16 + pchan) * 2;
        plz[b] = pch_ok ? p.Dp - lz : 0; ply[b] = p.Hp - ly; plx[b] = p.Wp - lx - pshift;
    }
    int qrel[NITQ], qlz[NITQ], qly[NITQ], qlx[NITQ];      // Q fragment b inside iff (unsigned)(qz0 + qlz[b]) < Dq && ... (qz0 >= -1)
#pragma unroll
    for (int b = 0; b < NITQ; ++b) {
        const int u = tid + b * 256;
        const int v = u / QU;
        const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
        qrel[b] = (((lz * p.Hq + ly) * p.Wq + lx) * p.Cch + cb * CB + qpart * 8) * 2;
        qlz[b] = (u < NQ && qch_ok) ? lz : 0x40000000; qly[b] = ly; qlx[b] = lx;
    }
    u32x4 pv[2], qv[NITQ];
    unsigned int okbits = 0;                     // bit b: P fragment b inside the volume; bit 2 + b: Q fragment b
    // Stride-2 kind, round 6: the bias gradient of a ConvTranspose3d is the per-channel sum of the very tensor this kernel stages as Q (its fine output gradient);
    // the m-block-0 workgroups add their Q fragments up on the way to LDS instead of a second pass re-reading the tensor (56.6 MB at 96^3, p.up_co = where the
    // partial sums go, in doubles behind p.ws; 0 = not asked for).  Out-of-volume fragments arrive as zeros from the bounds-checked loads.
    const bool qfold = KIND == G3_K2S2 && p.up_co != 0 && mb == 0;
    float qsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto request = [&](int t) {
        int n, z0, y0, x0;
        g3_tile_coords(p, t, n, z0, y0, x0);
        okbits = 0;
        const int pbase = KIND == G3_UP ? (((n * 2 * p.Dp + 2 * z0) * up_fh + 2 * y0) * up_fw + 2 * x0) * p.up_co * 2
                                        : (((n * p.Dp + z0) * p.Hp + y0) * p.Wp + x0) * p.Mch * 2;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool ok = z0 < plz[b] && y0 < ply[b] && x0 < plx[b];
            okbits |= ok ? (1u << b) : 0u;
            pv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(prsrc, ok ? pbase + prel[b] : -1, 0, 0));
        }
        const int qz0 = KIND != G3_K2S2 ? z0 - 1 : 2 * z0, qy0 = KIND != G3_K2S2 ? y0 - 1 : 2 * y0, qx0 = KIND != G3_K2S2 ? x0 - 1 : 2 * x0;
        const int qbase = (((n * p.Dq + qz0) * p.Hq + qy0) * p.Wq + qx0) * p.Cch * 2;
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const bool ok = (unsigned)(qz0 + qlz[b]) < (unsigned)p.Dq && (unsigned)(qy0 + qly[b]) < (unsigned)p.Hq && (unsigned)(qx0 + qlx[b]) < (unsigned)p.Wq;
            okbits |= ok ? (4u << b) : 0u;
            qv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(qrsrc, ok ? qbase + qrel[b] : -1, 0, 0));
        }
    };
    auto commit = [&](int n) {                   // registers -> (normalised) LDS tiles
        f32x2 sc[4], sh[4];
        const bool p_stats = p_stats_in && !(ablate & 4), q_stats = q_stats_in && !(ablate & 4);
        if (p_stats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_psc + n * 16 + pchan + 2 * i);
                sh[i] = *(const f32x2*)(s_psh + n * 16 + pchan + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            u32x4 v = pv[b];
            if (p_stats) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            *(u32x4*)(s_p + ((tid + b * 256) >> 1) * 32 + ppart * 16) = v;
        }
        if (q_stats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_qsc + n * 16 + qpart * 8 + 2 * i);
                sh[i] = *(const f32x2*)(s_qsh + n * 16 + qpart * 8 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            u32x4 v = qv[b];
            if constexpr (KIND == G3_K2S2) {
                if (qfold) {
                    float f[8];
                    frag_unpack(v, f, (T*)nullptr);
#pragma unroll
                    for (int j = 0; j < 8; ++j) qsum[j] += f[j];
                }
            }
            if (q_stats) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> (2 + b)) & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            const int u = tid + b * 256;
            if (b < NITQ - 1 || u < NQ) *(u32x4*)(s_q + (u / QU) * QROW + qpart * 16) = v;
        }
    };

    int t = ks;                                  // ksplit never exceeds the tile count
    request(t);
    for (int i = tid; i < p.N * 16; i += 256) {
        const int n = i >> 4, c = i & 15;
        float m = 0.f, r = 1.f;
        const int pc = mb * 16 + c;
        if (p_stats_in && pc < p.Mch) stats_to_mean_rstd_fast(p.P_stats, (size_t)n * p.Mch + pc, (size_t)p.N * p.Mch, p.inv_cnt_p, p.eps, m, r);
        s_psc[i] = r; s_psh[i] = -m * r;
        m = 0.f; r = 1.f;
        const int qc = cb * CB + c;
        if (q_stats_in && c < CB && qc < p.Cch) stats_to_mean_rstd_fast(p.Q_stats, (size_t)n * p.Cch + qc, (size_t)p.N * p.Cch, p.inv_cnt_q, p.eps, m, r);
        s_qsc[i] = r; s_qsh[i] = -m * r;
    }

    // per-lane byte offsets into the Q tile of this lane's tr-read row for each column block (tap part only)
    int qoff[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        int tap = CB == 16 ? k : 2 * k + (p4 >> 1);
        if (tap >= NTAPS) tap = KIND != G3_K2S2 ? 13 : 0;
        int dz, dy, dx;
        if (KIND != G3_K2S2) { dz = tap / 9; dy = (tap / 3) % 3; dx = tap % 3; }
        else { dz = (tap >> 2) & 1; dy = (tap >> 1) & 1; dx = tap & 1; }
        if (mp) { dz = k / 3; dy = k % 3; dx = 1 + (p4 >> 1); }          // block k = (dz, dy): columns 0..7 the centre tap in x, 8..15 the tap one voxel on
        qoff[k] = ((dz * QY + dy) * QX + dx) * QROW + (CB == 16 ? p4 * 8 : (p4 & 1) * 8);
    }
    const int nk = mp ? (CB == 16 ? 18 : 9) : NCB;

    f32x4 acc[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (; t < p.total_tiles; t += p.ksplit) {
        if (!(ablate & 32) || t == ks) __syncthreads();      // tables visible / every wave is done reading the previous tile
        if (!(ablate & 16)) commit(fdiv(t, p.fd_m[0], p.fd_s & 0xff));
        if (!(ablate & 32)) __syncthreads();
        if (t + p.ksplit < p.total_tiles && !(ablate & 8)) request(t + p.ksplit);
        if (ablate & 1) continue;
        // ---- two K-steps of 32 voxels: y rows (2s, 2s+1) of this wave's z-slice ----
        if constexpr (CB == 16 && KIND != G3_K2S2) {
            // 16-channel 3x3x3 blocks: tap (dz, dy, dx) of K-step s needs the Q rows (2s + dy, 2s + dy + 1) of plane wave + dz at x shift dx —
            // the six rows of a plane serve all (s, dy): each row fragment is read ONCE per (dz, dx) (18 transposing reads per plane) and the
            // B operands are register pairs of them, instead of 2 reads per MFMA (the kernel is bound by these reads: 112 -> 58 per tile and wave)
            const int xr = 4 * g + q4;
            u32x4 a2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pa0 = (((wave * 4 + 2 * s) * 16 + xr) * 32) + p4 * 8;
                a2[s] = tr_pair(s_p, pa0, pa0 + 16 * 32);
            }
            if (mp) {                            // M-packed rows: the taps dx = 0, +1 (halo columns 1, 2) only, block (dz, dy, dx select)
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    s16x4 rr[6][2];
#pragma unroll
                    for (int yr = 0; yr < 6; ++yr)
#pragma unroll
                        for (int dxs = 0; dxs < 2; ++dxs)
                            rr[yr][dxs] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_q + (((wave + dz) * QY + yr) * QX + xr + 1 + dxs) * QROW + p4 * 8));
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dxs = 0; dxs < 2; ++dxs) {
                                typedef __attribute__((ext_vector_type(8))) short s16x8;
                                const s16x8 v = __builtin_shufflevector(rr[2 * s + dy][dxs], rr[2 * s + dy + 1][dxs], 0, 1, 2, 3, 4, 5, 6, 7);
                                const int k = (dz * 3 + dy) * 2 + dxs;
                                acc[k] = mfma16(a2[s], __builtin_bit_cast(u32x4, v), acc[k], (T*)nullptr);
                            }
                }
            } else
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                s16x4 rr[6][3];
#pragma unroll
                for (int yr = 0; yr < 6; ++yr)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        rr[yr][dx] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_q + (((wave + dz) * QY + yr) * QX + xr + dx) * QROW + p4 * 8));
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            typedef __attribute__((ext_vector_type(8))) short s16x8;
                            const s16x8 v = __builtin_shufflevector(rr[2 * s + dy][dx], rr[2 * s + dy + 1][dx], 0, 1, 2, 3, 4, 5, 6, 7);
                            const int k = dz * 9 + dy * 3 + dx;
                            acc[k] = mfma16(a2[s], __builtin_bit_cast(u32x4, v), acc[k], (T*)nullptr);
                        }
            }
        } else
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int xr = 4 * g + q4;                                   // this lane's tr-read row: voxel x
            const int pa0 = (((wave * 4 + 2 * s) * 16 + xr) * 32) + p4 * 8;
            const u32x4 a = tr_pair(s_p, pa0, pa0 + 16 * 32);
            int qb0, qb1;
            if (KIND != G3_K2S2) {
                qb0 = ((wave * QY + 2 * s) * QX + xr) * QROW;
                qb1 = qb0 + QX * QROW;
            } else {
                qb0 = ((2 * wave * QY + 2 * (2 * s)) * QX + 2 * xr) * QROW;
                qb1 = qb0 + 2 * QX * QROW;
            }
            constexpr int NK0 = (CB == 8 && KIND == G3_K3) ? 9 : NCB;      // the M-packed form stops after 9 blocks
#pragma unroll
            for (int k = 0; k < NK0; ++k) {
                const u32x4 b = tr_pair(s_q, qb0 + qoff[k], qb1 + qoff[k]);
                acc[k] = mfma16(a, b, acc[k], (T*)nullptr);
            }
            if (!mp) {
#pragma unroll
                for (int k = NK0; k < NCB; ++k) {
                    const u32x4 b = tr_pair(s_q, qb0 + qoff[k], qb1 + qoff[k]);
                    acc[k] = mfma16(a, b, acc[k], (T*)nullptr);
                }
            }
        }
    }

    const size_t slab_elems = (size_t)p.mbn * p.cbn * nk * 256;
    g3_finish<NCB>(acc, (float*)s_p, p.ws + (size_t)ks * slab_elems + ((size_t)bx * nk) * 256, wave, col, g, nk);
    if constexpr (KIND == G3_K2S2) {
        if (qfold) {                             // uniform per workgroup: partial sums [k-split][Cch] in double, summed by g3_reduce_group_kernel like bias_partial_body's
            float* s_red = (float*)s_q;          // every wave left the tile loop before g3_finish's barriers
#pragma unroll
            for (int j = 0; j < 8; ++j) s_red[tid * 8 + j] = qsum[j];
            __syncthreads();
            if (tid < CB && cb * CB + tid < p.Cch) {
                const int part = tid >> 3, j = tid & 7;
                double tot = 0.0;
                for (int y = 0; y < 256 / QU; ++y) tot += (double)s_red[(y * QU + part) * 8 + j];
                ((double*)p.ws + (size_t)p.up_co)[(size_t)ks * p.Cch + cb * CB + tid] = tot;
            }
        }
    }
}

template <int CB, int KIND, typename T = unsigned short>
__global__ __launch_bounds__(256) void g3b_kernel(const G3Params p) { g3b_body<T, CB, KIND>(p, blockIdx.x, blockIdx.y); }

// ---------------------------------------------------------------------------------------------------
// fp32 parity mode, 3x3x3 layers: the same GEMM on the bf16 matrix cores through three-limb operand splitting (igemm_k3x.h explains the
// arithmetic: x = x0 + x1 + x2 in bf16 limbs, six exact limb products per product, fp32 accumulation — 2.7x fewer matrix cycles than
// g3_kernel's exact-f32 MFMAs).  Structure of g3b_body: P and Q tiles are fetched one tile ahead (bounds-checked buffer loads of fp32
// fragments, 4 channels each), normalised in fp32, split, and written to LDS as three limb PLANES of the bf16 tile layout; the operands are
// read with ds_read_b64_tr_b16 per plane.  Slab layout and reduction are g3b_kernel's.  (Stride-2 kinds keep g3_kernel: their three Q limb
// planes of 8x8x32 voxels would not fit the LDS.)
// ---------------------------------------------------------------------------------------------------
#define G3X_LDS_P (4 * G3_MAXN * 16 * 4)
#define G3X_LDS_Q (G3X_LDS_P + 3 * 256 * 16 * 2)

template <int CB>
__device__ __forceinline__ void g3x_body(const G3Params& p, const int bx, const int ks) {
    using GEO = G3Geo<CB, G3_K3>;
    constexpr int NCB = GEO::NCB, QY = GEO::QY, QX = GEO::QX, QV = GEO::QV;
    constexpr int QROW = CB * 2;                 // bytes per Q-tile voxel in one limb plane
    constexpr int QU = CB / 4;                   // fp32 fragments (4 channels) per Q-tile voxel
    constexpr int NQ = QV * QU;
    constexpr int NITQ = (NQ + 255) / 256;
    constexpr int PPB = 256 * 32, QPB = QV * QROW;      // bytes of one P / Q limb plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_pm = (float*)(smem + G3_LDS_STATS);        // mean / rstd of P's and Q's 16 channels, per sample (the fp32 mode's (x - mean) * rstd)
    float* s_pr = s_pm + G3_MAXN * 16;
    float* s_qm = s_pr + G3_MAXN * 16;
    float* s_qr = s_qm + G3_MAXN * 16;
    char* s_p = smem + G3X_LDS_P;                // [3 limbs][256 voxels][16 ch] bf16
    char* s_q = smem + G3X_LDS_Q;                // [3 limbs][QV voxels][CB ch] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int q4 = col >> 2, p4 = col & 3;
    const int mb = bx / p.cbn, cb = bx - mb * p.cbn;
    const bool p_stats = p.P_stats != nullptr, q_stats = p.Q_stats != nullptr;
    const i32x4 prsrc = make_rsrc(p.P, (unsigned int)((long long)p.N * p.Dp * p.Hp * p.Wp * p.Mch * 4));
    const i32x4 qrsrc = make_rsrc(p.Q, (unsigned int)((long long)p.N * p.Dq * p.Hq * p.Wq * p.Cch * 4));

    // P fragment b: voxel (tid + 256 b) >> 2 of the 4x4x16 tile, channels 4 * ppart ..; Q fragment b: voxel u / QU of the halo region, channels 4 * qpart ..
    const int ppart = tid & 3, qpart = tid % QU;
    const bool pch_ok = mb * 16 + ppart * 4 < p.Mch, qch_ok = cb * CB + qpart * 4 < p.Cch;
    int prel[4], pzyx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int v = (tid + b * 256) >> 2;
        const int lx = v & 15, ly = (v >> 4) & 3, lz = v >> 6;
        prel[b] = (((lz * p.Hp + ly) * p.Wp + lx) * p.Mch + mb * 16 + ppart * 4) * 4;
        pzyx[b] = pch_ok ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    int qrel[NITQ], qzyx[NITQ];
#pragma unroll
    for (int b = 0; b < NITQ; ++b) {
        const int u = tid + b * 256;
        const int v = u / QU;
        const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
        qrel[b] = (((lz * p.Hq + ly) * p.Wq + lx) * p.Cch + cb * CB + qpart * 4) * 4;
        qzyx[b] = (u < NQ && qch_ok) ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    u32x4 pv[4], qv[NITQ];
    unsigned int okbits = 0;                     // bit b: P fragment b inside the volume; bit 4 + b: Q fragment b
    auto request = [&](int t) {
        int n, z0, y0, x0;
        g3_tile_coords(p, t, n, z0, y0, x0);
        okbits = 0;
        const int pbase = (((n * p.Dp + z0) * p.Hp + y0) * p.Wp + x0) * p.Mch * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gz = z0 + (pzyx[b] & 0xff), gy = y0 + ((pzyx[b] >> 8) & 0xff), gx = x0 + (pzyx[b] >> 16);
            const bool ok = gz < p.Dp && gy < p.Hp && gx < p.Wp;
            okbits |= ok ? (1u << b) : 0u;
            pv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(prsrc, ok ? pbase + prel[b] : -1, 0, 0));
        }
        const int qz0 = z0 - 1, qy0 = y0 - 1, qx0 = x0 - 1;
        const int qbase = (((n * p.Dq + qz0) * p.Hq + qy0) * p.Wq + qx0) * p.Cch * 4;
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const int gz = qz0 + (qzyx[b] & 0xff), gy = qy0 + ((qzyx[b] >> 8) & 0xff), gx = qx0 + (qzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.Dq && (unsigned)gy < (unsigned)p.Hq && (unsigned)gx < (unsigned)p.Wq;
            okbits |= ok ? (16u << b) : 0u;
            qv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(qrsrc, ok ? qbase + qrel[b] : -1, 0, 0));
        }
    };
    auto put = [&](const u32x4 raw, bool lazy, bool ok, const float* s_m, const float* s_r, char* plane0, int plane_bytes) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(raw[j]);
        if (lazy) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (v[j] - s_m[j]) * s_r[j];
                v[j] = ok ? fmaxf(t, 0.f) : 0.f;          // zero padding applies to the normalised activation
            }
        }
        unsigned int lm[3][2];
        vs_limb_split4(v, lm);
#pragma unroll
        for (int l = 0; l < 3; ++l) *(u32x2*)(plane0 + l * plane_bytes) = u32x2{lm[l][0], lm[l][1]};
    };
    auto commit = [&](int n) {                   // registers -> normalised limb planes
#pragma unroll
        for (int b = 0; b < 4; ++b)
            put(pv[b], p_stats, (okbits >> b) & 1u, s_pm + n * 16 + ppart * 4, s_pr + n * 16 + ppart * 4, s_p + ((tid + b * 256) >> 2) * 32 + ppart * 8, PPB);
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const int u = tid + b * 256;
            if (b < NITQ - 1 || u < NQ)
                put(qv[b], q_stats, (okbits >> (4 + b)) & 1u, s_qm + n * 16 + qpart * 4, s_qr + n * 16 + qpart * 4, s_q + (u / QU) * QROW + qpart * 8, QPB);
        }
    };

    int t = ks;                                  // ksplit never exceeds the tile count
    request(t);
    for (int i = tid; i < p.N * 16; i += 256) {
        const int n = i >> 4, c = i & 15;
        float m = 0.f, r = 1.f;
        const int pc = mb * 16 + c;
        if (p_stats && pc < p.Mch) stats_to_mean_rstd(p.P_stats, (size_t)n * p.Mch + pc, (size_t)p.N * p.Mch, p.inv_cnt_p, p.eps, m, r);
        s_pm[i] = m; s_pr[i] = r;
        m = 0.f; r = 1.f;
        const int qc = cb * CB + c;
        if (q_stats && c < CB && qc < p.Cch) stats_to_mean_rstd(p.Q_stats, (size_t)n * p.Cch + qc, (size_t)p.N * p.Cch, p.inv_cnt_q, p.eps, m, r);
        s_qm[i] = m; s_qr[i] = r;
    }
    // per-lane byte offsets into a Q limb plane of this lane's tr-read row for each column block (tap part only)
    int qoff[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        int tap = CB == 16 ? k : 2 * k + (p4 >> 1);
        if (tap >= 27) tap = 13;
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        qoff[k] = ((dz * QY + dy) * QX + dx) * QROW + (CB == 16 ? p4 * 8 : (p4 & 1) * 8);
    }
    f32x4 acc[NCB], acl[NCB];                    // leading products / the five small limb products (igemm_k3x.h: two accumulators)
#pragma unroll
    for (int k = 0; k < NCB; ++k) { acc[k] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (; t < p.total_tiles; t += p.ksplit) {
        __syncthreads();                         // tables visible / every wave is done reading the previous tile
        commit(fdiv(t, p.fd_m[0], p.fd_s & 0xff));
        __syncthreads();
        if (t + p.ksplit < p.total_tiles) request(t + p.ksplit);
        // two K-steps of 32 voxels: y rows (2s, 2s+1) of this wave's z-slice; per column block three B limbs against three A limbs, six MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int xr = 4 * g + q4;                                   // this lane's tr-read row: voxel x
            const int pa0 = (((wave * 4 + 2 * s) * 16 + xr) * 32) + p4 * 8;
            u32x4 a[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) a[l] = tr_pair(s_p + l * PPB, pa0, pa0 + 16 * 32);
            const int qb0 = ((wave * QY + 2 * s) * QX + xr) * QROW;
            const int qb1 = qb0 + QX * QROW;
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                u32x4 b[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) b[l] = tr_pair(s_q + l * QPB, qb0 + qoff[k], qb1 + qoff[k]);
                // limb pairs (P limb i, Q limb j), smallest products first
                acl[k] = mfma16(a[2], b[0], acl[k], (unsigned short*)nullptr);
                acl[k] = mfma16(a[1], b[1], acl[k], (unsigned short*)nullptr);
                acl[k] = mfma16(a[0], b[2], acl[k], (unsigned short*)nullptr);
                acl[k] = mfma16(a[1], b[0], acl[k], (unsigned short*)nullptr);
                acl[k] = mfma16(a[0], b[1], acl[k], (unsigned short*)nullptr);
                acc[k] = mfma16(a[0], b[0], acc[k], (unsigned short*)nullptr);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] += acl[k];
    const size_t slab_elems = (size_t)p.mbn * p.cbn * NCB * 256;
    g3_finish<NCB>(acc, (float*)s_p, p.ws + (size_t)ks * slab_elems + ((size_t)bx * NCB) * 256, wave, col, g);
}

template <int CB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CB == 8 ? 2 : 1))) void g3x_kernel(const G3Params p) { g3x_body<CB>(p, blockIdx.x, blockIdx.y); }


// ---------------------------------------------------------------------------------------------------
// Big-tile 3x3x3 weight gradients (round 5): g3c_body.
// The grouped launches move 2.15 x their algorithmic bytes (profiles/r04_hbm_traffic.txt) at ~4.6 TB/s: they are bound by that traffic, not by
// the instruction stream (the operand exchange of multi_plan removed a third of their vector instructions for 0.5 % of the step).  The excess
// is the halo: a 4x4x16 tile stages (6 x 6 x 18) / (4 x 4 x 16) = 2.53 x its voxels of the halo operand.  Here a tile is TZ x TY x 16 = 8 x 8 x 16
// voxels — halo 10 x 10 x 18 = 1.76 x — for the layers whose tensors are big enough to matter (the full-resolution ones).  Operands:
//   A ("P", rows of the MFMA)   the centre operand, no halo: 16 channels of a voxel, or (MP) 8 channels x {this voxel, the next one in x}: the
//                               shifted rows are READ from the tile one voxel on (a 17-wide tile), not staged twice as g3b_body's M-packed form does
//   B ("Q", column blocks)      the halo operand, CBH = 8 channels per voxel: blocks of two taps (14), or (MP) of (dz, dy) x {dx = 0, +1} (9)
// Slab layout, statistics tables, tile walk (ks, ks + ksplit, ...) and the reduction are g3b_body's.
template <int CBH, bool MP, int TZ, int TY> struct G3CGeo {
    static constexpr int AC = MP ? 8 : 16, XW = MP ? 17 : 16, AU = AC / 8;
    static constexpr int PV = TZ * TY * XW, NP = PV * AU, NITP = (NP + 255) / 256;
    static constexpr int QZ = TZ + 2, QY = TY + 2, QX = 18, QV = QZ * QY * QX, QU = CBH / 8, NQ = QV * QU, NITQ = (NQ + 255) / 256;
    static constexpr int NCB = MP ? (CBH == 16 ? 18 : 9) : (CBH == 16 ? 27 : 14);
    static constexpr int P_BYTES = NITP * 4096, Q_BYTES = NITQ * 4096;             // padded: every thread stores every fragment it holds
    static constexpr size_t LDS = G3B_LDS_P + P_BYTES + Q_BYTES;
};

template <typename T, int CBH, bool MP, int TZ, int TY>
__device__ __forceinline__ void g3c_body(const G3Params& p, const int bx, const int ks) {
    using GEO = G3CGeo<CBH, MP, TZ, TY>;
    constexpr int AC = GEO::AC, XW = GEO::XW, AU = GEO::AU, NP = GEO::NP, NITP = GEO::NITP;
    constexpr int QY = GEO::QY, QX = GEO::QX, QU = GEO::QU, NQ = GEO::NQ, NITQ = GEO::NITQ, NCB = GEO::NCB;
    constexpr int AROW = AC * 2, QROW = CBH * 2;
    static_assert(TZ % 4 == 0 && TY % 2 == 0, "a wave takes TZ / 4 z-slices of TY / 2 row pairs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_psc = (float*)(smem + G3_LDS_STATS);
    float* s_psh = s_psc + G3_MAXN * 16;
    float* s_qsc = s_psh + G3_MAXN * 16;
    float* s_qsh = s_qsc + G3_MAXN * 16;
    char* s_p = smem + G3B_LDS_P;
    char* s_q = s_p + GEO::P_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int q4 = col >> 2, p4 = col & 3;
    const int mb = bx / p.cbn, cb = bx - mb * p.cbn;
    const bool p_stats_in = p.P_stats != nullptr, q_stats_in = p.Q_stats != nullptr;
    const i32x4 prsrc = make_rsrc(p.P, (unsigned int)((long long)p.N * p.Dp * p.Hp * p.Wp * p.Mch * 2));
    const i32x4 qrsrc = make_rsrc(p.Q, (unsigned int)((long long)p.N * p.Dq * p.Hq * p.Wq * p.Cch * 2));

    // ---- tile-independent fragment geometry (limit form, as g3b_body) ----
    const int ppart = tid % AU, qpart = tid % QU;          // 256 % AU == 0, 256 % QU == 0: constant per thread
    const bool pch_ok = (MP ? 0 : mb * 16) + ppart * 8 < p.Mch, qch_ok = cb * CBH + qpart * 8 < p.Cch;
    // local coordinates packed (x | y << 8 | z << 16; z = 255 for a fragment that must read as zero): the kernel is bound by its traffic, not by the unpacking,
    // and 16 fragments x 4 registers of geometry would not fit two waves per SIMD next to the accumulators
    int prel[NITP], pco[NITP];
#pragma unroll
    for (int b = 0; b < NITP; ++b) {
        const int u = tid + b * 256, v = u / AU;
        const int lx = v % XW, ly = (v / XW) % TY, lz = v / (XW * TY);
        prel[b] = (((lz * p.Hp + ly) * p.Wp + lx) * p.Mch + (MP ? 0 : mb * 16) + ppart * 8) * 2;
        pco[b] = (u < NP && pch_ok) ? (lx | (ly << 8) | (lz << 16)) : 0x7fffffff;
    }
    int qrel[NITQ], qco[NITQ];
#pragma unroll
    for (int b = 0; b < NITQ; ++b) {
        const int u = tid + b * 256, v = u / QU;
        const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
        qrel[b] = (((lz * p.Hq + ly) * p.Wq + lx) * p.Cch + cb * CBH + qpart * 8) * 2;
        qco[b] = (u < NQ && qch_ok) ? (lx | (ly << 8) | (lz << 16)) : 0x7fffffff;
    }
    u32x4 pv[NITP], qv[NITQ];
    unsigned int okbits = 0;                              // bit b: P fragment b inside the volume; bit 8 + b: Q fragment b
    auto coords = [&](int t, int& n, int& z0, int& y0, int& x0) {
        n = fdiv(t, p.fd_m[0], p.fd_s & 0xff);
        const int tl = t - n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], (p.fd_s >> 8) & 0xff);
        const int r = tl - tz * p.txn * p.tyn;
        const int ty = fdiv(r, p.fd_m[2], (p.fd_s >> 16) & 0xff);
        z0 = tz * TZ; y0 = ty * TY; x0 = (r - ty * p.txn) * 16;
    };
    auto request = [&](int t) {
        int n, z0, y0, x0;
        coords(t, n, z0, y0, x0);
        okbits = 0;
        const int pbase = (((n * p.Dp + z0) * p.Hp + y0) * p.Wp + x0) * p.Mch * 2;
#pragma unroll
        for (int b = 0; b < NITP; ++b) {
            const bool ok = z0 + (pco[b] >> 16) < p.Dp && y0 + ((pco[b] >> 8) & 0xff) < p.Hp && x0 + (pco[b] & 0xff) < p.Wp;
            okbits |= ok ? (1u << b) : 0u;
            pv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(prsrc, ok ? pbase + prel[b] : -1, 0, 0));
        }
        const int qz0 = z0 - 1, qy0 = y0 - 1, qx0 = x0 - 1;
        const int qbase = (((n * p.Dq + qz0) * p.Hq + qy0) * p.Wq + qx0) * p.Cch * 2;
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const bool ok = (unsigned)(qz0 + (qco[b] >> 16)) < (unsigned)p.Dq && (unsigned)(qy0 + ((qco[b] >> 8) & 0xff)) < (unsigned)p.Hq &&
                            (unsigned)(qx0 + (qco[b] & 0xff)) < (unsigned)p.Wq;
            okbits |= ok ? (0x100u << b) : 0u;
            qv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(qrsrc, ok ? qbase + qrel[b] : -1, 0, 0));
        }
    };
    auto commit = [&](int n) {                             // registers -> (normalised) LDS tiles; fragment u lies at byte 16 u of its tile
        f32x2 sc[4], sh[4];
        if (p_stats_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_psc + n * 16 + ppart * 8 + 2 * i);
                sh[i] = *(const f32x2*)(s_psh + n * 16 + ppart * 8 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NITP; ++b) {
            u32x4 v = pv[b];
            if (p_stats_in) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            *(u32x4*)(s_p + (tid + b * 256) * 16) = v;
        }
        if (q_stats_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_qsc + n * 16 + qpart * 8 + 2 * i);
                sh[i] = *(const f32x2*)(s_qsh + n * 16 + qpart * 8 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            u32x4 v = qv[b];
            if (q_stats_in) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> (8 + b)) & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            *(u32x4*)(s_q + (tid + b * 256) * 16) = v;
        }
    };

    int t = ks;
    request(t);
    for (int i = tid; i < p.N * 16; i += 256) {
        const int n = i >> 4, c = i & 15;
        float m = 0.f, r = 1.f;
        const int pc = (MP ? 0 : mb * 16) + c;
        if (p_stats_in && c < AC && pc < p.Mch) stats_to_mean_rstd_fast(p.P_stats, (size_t)n * p.Mch + pc, (size_t)p.N * p.Mch, p.inv_cnt_p, p.eps, m, r);
        s_psc[i] = r; s_psh[i] = -m * r;
        m = 0.f; r = 1.f;
        const int qc = cb * CBH + c;
        if (q_stats_in && c < CBH && qc < p.Cch) stats_to_mean_rstd_fast(p.Q_stats, (size_t)n * p.Cch + qc, (size_t)p.N * p.Cch, p.inv_cnt_q, p.eps, m, r);
        s_qsc[i] = r; s_qsh[i] = -m * r;
    }

    // this lane's transposing-read offsets: A relative to (z-slice, row 2s [+1], x = 0), B per column block relative to the halo voxel of the same (z, y, x)
    const int xr = 4 * g + q4;
    const int pa_lane = MP ? ((xr + (p4 >> 1)) * AROW + (p4 & 1) * 8) : (xr * AROW + p4 * 8);
    int qoff[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        int dz, dy, dx, choff;
        if (MP) {
            if (CBH == 16) { dz = (k >> 1) / 3; dy = (k >> 1) % 3; dx = 1 + (k & 1); choff = p4 * 8; }
            else { dz = k / 3; dy = k % 3; dx = 1 + (p4 >> 1); choff = (p4 & 1) * 8; }
        } else {
            int tap = CBH == 16 ? k : 2 * k + (p4 >> 1);
            if (tap >= 27) tap = 13;
            dz = tap / 9; dy = (tap / 3) % 3; dx = tap % 3;
            choff = CBH == 16 ? p4 * 8 : (p4 & 1) * 8;
        }
        qoff[k] = ((dz * QY + dy) * QX + dx + xr) * QROW + choff;
    }

    f32x4 acc[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (; t < p.total_tiles; t += p.ksplit) {
        __syncthreads();                                   // tables visible / every wave is done reading the previous tile
        commit(fdiv(t, p.fd_m[0], p.fd_s & 0xff));
        __syncthreads();
        if (t + p.ksplit < p.total_tiles) request(t + p.ksplit);
#pragma unroll
        for (int zi = 0; zi < TZ / 4; ++zi) {
            const int zs = wave * (TZ / 4) + zi;
#pragma unroll
            for (int s = 0; s < TY / 2; ++s) {
                const int pa0 = ((zs * TY + 2 * s) * XW) * AROW + pa_lane;
                const u32x4 a = tr_pair(s_p, pa0, pa0 + XW * AROW);
                const int qb0 = ((zs * QY + 2 * s) * QX) * QROW;
#pragma unroll
                for (int k = 0; k < NCB; ++k) {
                    const u32x4 b = tr_pair(s_q, qb0 + qoff[k], qb0 + QX * QROW + qoff[k]);
                    acc[k] = mfma16(a, b, acc[k], (T*)nullptr);
                }
            }
        }
    }
    __syncthreads();
    const size_t slab_elems = (size_t)p.mbn * p.cbn * NCB * 256;
    g3_finish<NCB>(acc, (float*)s_p, p.ws + (size_t)ks * slab_elems + ((size_t)bx * NCB) * 256, wave, col, g, NCB);
}

// Grouped launch: the weight gradients of up to G3_GROUP_MAX layers of one (CB, KIND) instantiation in ONE grid.  Weight
// gradients are leaves of backward, so the host defers them to the end of the pass and issues them together: the small
// layers (tens of workgroups each, start-up bound) then share the chip instead of queueing behind one another.
// Workgroup b belongs to layer l with wg_start[l] <= b < wg_start[l+1]; inside the layer it is (pair, k-split) as in g3b_kernel.
#define G3_GROUP_MAX 24
struct G3Group {
    G3Params p[G3_GROUP_MAX];
    int wg_start[G3_GROUP_MAX + 1];
    int n;
    int xcd;          // g3_xcd_rank below: 0 = plain order, 1 = single-block-pair layers only (rounds 3-5), 2 = every layer (default since round 6); VS_WGRAD_XCD
};

static_assert(sizeof(G3Group) <= 4096, "G3Group travels as the kernel argument");

// The hardware deals consecutive workgroup ids to the 8 XCDs in turn.  A layer's workgroups [b0, b0 + count) are re-ranked so that XCD x owns ONE contiguous run of
// ranks; rank -> (k-split = rank / pairs, block pair = rank % pairs).  The k-splits of a run walk neighbouring tiles (their Q halos meet in that XCD's L2) and, round 6,
// the block pairs of one k-split — mbn x cbn workgroups that stage the SAME P / Q tiles for different channel blocks — sit on one XCD too instead of on mbn * cbn
// different ones (each with an L2 of its own: a 64 -> 32 layer at 24^3 fetched its tensors 7.7 x, profiles/r06_wgrad_layer_traffic.txt).
__device__ __forceinline__ int g3_xcd_rank(int b0, int local, int count, bool on) {
    if (!on) return local;
    const int x = (b0 + local) & 7;
    int start = 0;
#pragma unroll
    for (int xx = 0; xx < 8; ++xx) {
        const int first = (xx - b0) & 7;                                        // smallest local index on XCD xx
        const int cnt = first < count ? (count - first + 7) >> 3 : 0;
        start += xx < x ? cnt : 0;
    }
    return start + ((local - ((x - b0) & 7)) >> 3);
}

template <int CB, int KIND, typename T = unsigned short, bool MP = false>
// (3 waves per SIMD for the 8-channel bucket — the compiler gets there without AGPRs — changed nothing: 2.814 vs 2.810 ms per step)
__global__ __launch_bounds__(256) void g3b_group_kernel(const G3Group grp) {
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_GROUP_MAX; ++i) l += (i < grp.n && b >= grp.wg_start[i]) ? 1 : 0;
    const G3Params p = grp.p[l];
    const int local = b - grp.wg_start[l];
    const int pairs = p.mbn * p.cbn;
    const int rank = (grp.xcd && p.ksplit * pairs >= 16) ? g3_xcd_rank(grp.wg_start[l], local, p.ksplit * pairs, grp.xcd > 1 || pairs == 1) : local;
    const int ks = rank / pairs;
    g3b_body<T, CB, KIND, MP>(p, rank - ks * pairs, ks);
}

// the same grouping for the limb kernel of the fp32 parity mode (3x3x3 layers): one grid per channel-block width
template <int CB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CB == 8 ? 2 : 1))) void g3x_group_kernel(const G3Group grp) {
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_GROUP_MAX; ++i) l += (i < grp.n && b >= grp.wg_start[i]) ? 1 : 0;
    const G3Params p = grp.p[l];
    const int local = b - grp.wg_start[l];
    const int pairs = p.mbn * p.cbn;
    const int rank = (grp.xcd > 1 && p.ksplit * pairs >= 16) ? g3_xcd_rank(grp.wg_start[l], local, p.ksplit * pairs, true) : local;
    g3x_body<CB>(p, rank - (rank / pairs) * pairs, rank / pairs);
}

// Sum the partial slabs (fixed order, fp64) into the reference's [m][c][tap] layout.  A block = 64 consecutive slab
// elements (coalesced reads) x 16 slab partitions; each thread keeps 8 loads in flight, LDS combines the partitions.
// fp32 mode, stride-2 kinds: the layers of a pass in one grid on the exact-f32 MFMA body (their Q limb planes would not fit the LDS)
template <int CB, int KIND>
__global__ __launch_bounds__(256) void g3_group_kernel(const G3Group grp) {
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_GROUP_MAX; ++i) l += (i < grp.n && b >= grp.wg_start[i]) ? 1 : 0;
    const G3Params p = grp.p[l];
    const int local = b - grp.wg_start[l];
    const int pairs = p.mbn * p.cbn;
    const int rank = (grp.xcd > 1 && p.ksplit * pairs >= 16) ? g3_xcd_rank(grp.wg_start[l], local, p.ksplit * pairs, true) : local;
    const int ks = rank / pairs;
    g3_body<float, CB, KIND>(p, rank - ks * pairs, ks);
}

template <int CB, int KIND>
__global__ __launch_bounds__(1024) void g3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int m_real, int c_real, int mbn,
                                                         int cbn, int nslabs) {
    using GEO = G3Geo<CB, KIND>;
    constexpr int NTAPS = GEO::NTAPS, NCB = GEO::NCB;
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const size_t slab_elems = (size_t)mbn * cbn * NCB * 256;
    const size_t e = (size_t)blockIdx.x * 64 + lane;          // element within a slab: [(mb*cbn+cb)][k][col][row]
    double s = 0.0;
    if (e < slab_elems) {
        const float* src = ws + e;
        int sl = part;
        for (; sl + 112 < nslabs; sl += 128) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(sl + 16 * j) * slab_elems];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; sl < nslabs; sl += 16) s += (double)src[(size_t)sl * slab_elems];
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0 && e < slab_elems) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][lane];
        const int row = (int)(e & 15), col = (int)((e >> 4) & 15);
        const int k = (int)((e >> 8) % NCB);
        const int pair = (int)(e / ((size_t)NCB * 256));
        const int mb = pair / cbn, cb = pair - mb * cbn;
        const int m = mb * 16 + row;
        int c, tap;
        if (CB == 16) { c = cb * 16 + col; tap = k; }
        else { c = cb * 8 + (col & 7); tap = 2 * k + (col >> 3); }
        if (m < m_real && c < c_real && tap < NTAPS) dw[((size_t)m * c_real + c) * NTAPS + tap] = (float)tot;
    }
}

static void g3_plan(int n, int dp, int hp, int wp, int m_ch, int c_ch, int kind, int& cbsz, int& mbn, int& cbn,
                    int& ncb, int& tiles_per_sample, int& tyn, int& txn, int& ksplit, bool short_chains = false) {
    cbsz = c_ch >= 16 ? 16 : 8;
    mbn = (m_ch + 15) / 16;
    cbn = (c_ch + cbsz - 1) / cbsz;
    const int ntaps = kind != VS_CONV_K2S2 ? 27 : 8;
    ncb = cbsz == 16 ? ntaps : (ntaps + 1) / 2;
    tyn = (hp + 3) / 4; txn = (wp + 15) / 16;
    tiles_per_sample = ((dp + 3) / 4) * tyn * txn;
    const long long total = (long long)tiles_per_sample * n;
    const long long wg_target = vs_cfg().wgrad_wgs;   // tuning knob: ~2 workgroups per CU overall
    long long want = (wg_target + (long long)mbn * cbn - 1) / ((long long)mbn * cbn);
    // fp32 (parity) mode: a bounded number of tiles per workgroup, so that an fp32 MFMA accumulator chains a bounded number of products (64 per
    // tile and wave) before the fp64 slab reduction: VS_WGRAD_F32_TILES tiles (see the default's comment)
    const long long f32_tiles = vs_cfg().wgrad_f32_tiles;      // 8 x 64 voxels x 32-wide MFMAs: 512-product chains per accumulator and limb pair
    if (short_chains && (total + f32_tiles - 1) / f32_tiles > want) want = (total + f32_tiles - 1) / f32_tiles;
    if (want < 1) want = 1;
    if (want > total) want = total;
    // keep the slab workspace <= 64 MiB
    const double slab_bytes = (double)mbn * cbn * ncb * 256 * 4;
    while (want > 1 && want * slab_bytes > 64.0 * 1024 * 1024) --want;
    ksplit = (int)want;
}

extern "C" size_t vs_conv_wgrad_workspace_bytes(int n, int dp, int hp, int wp, int m_ch, int c_ch, int kind) {
    // sized for the fp32-mode plan (the larger of the two), so one query serves both dtypes
    int cbsz, mbn, cbn, ncb, tps, tyn, txn, ksplit;
    g3_plan(n, dp, hp, wp, m_ch, c_ch, kind == VS_CONV_K3 ? VS_CONV_K3 : VS_CONV_K2S2, cbsz, mbn, cbn, ncb, tps, tyn, txn, ksplit, true);
    return (size_t)ksplit * mbn * cbn * ncb * 256 * 4;
}

// reduce_slabs: how many slabs (starting at reduce_ws) the reduction that follows this layer's kernel sums; 0 = no reduction launch (a
// later part of the same weight's gradient — a weight used several times in one backward pass — reduces all parts' slabs together)
template <int CB, int KIND>
static int g3_reduce_launch(const float* reduce_ws, float* dw, int m_real, int c_real, int mbn, int cbn, int slabs, hipStream_t s) {
    using GEO = G3Geo<CB, KIND>;
    const long long slab_elems = (long long)mbn * cbn * GEO::NCB * 256;
    hipLaunchKernelGGL((g3_reduce_kernel<CB, KIND>), dim3(vs_ceil_div(slab_elems, 64)), dim3(1024), 0, s, reduce_ws, dw, m_real, c_real, mbn, cbn, slabs);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T, int CB, int KIND>
static int g3b_run(const G3Params& p, float* dw, int m_real, int c_real, hipStream_t s, const float* reduce_ws, int reduce_slabs) {
    using GEO = G3Geo<CB, KIND>;
    constexpr size_t lds = G3B_LDS_Q + (size_t)GEO::QV * CB * 2;
    auto kern = g3b_kernel<CB, KIND, T>;
    if (lds > 64 * 1024) {
        static const hipError_t attr_err =
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr_err != hipSuccess) return (int)attr_err;
    }
    hipLaunchKernelGGL(kern, dim3(p.mbn * p.cbn, p.ksplit), dim3(256), lds, s, p);
    VS_CHECK_LAUNCH();
    return reduce_slabs > 0 ? g3_reduce_launch<CB, KIND>(reduce_ws, dw, m_real, c_real, p.mbn, p.cbn, reduce_slabs, s) : VS_OK;
}

template <typename T, int CB, int KIND>
static int g3_run(const G3Params& p, float* dw, int m_real, int c_real, hipStream_t s, const float* reduce_ws, int reduce_slabs) {
    using GEO = G3Geo<CB, KIND>;
    constexpr size_t lds = G3_LDS_Q + (size_t)GEO::QV * CB * 4;
    auto kern = g3_kernel<T, CB, KIND>;
    if (lds > 64 * 1024) {
        static const hipError_t attr_err =
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr_err != hipSuccess) return (int)attr_err;
    }
    hipLaunchKernelGGL(kern, dim3(p.mbn * p.cbn, p.ksplit), dim3(256), lds, s, p);
    VS_CHECK_LAUNCH();
    return reduce_slabs > 0 ? g3_reduce_launch<CB, KIND>(reduce_ws, dw, m_real, c_real, p.mbn, p.cbn, reduce_slabs, s) : VS_OK;
}

template <int CB>
static int g3x_run(const G3Params& p, float* dw, int m_real, int c_real, hipStream_t s, const float* reduce_ws, int reduce_slabs) {
    using GEO = G3Geo<CB, G3_K3>;
    constexpr size_t lds = G3X_LDS_Q + (size_t)3 * GEO::QV * CB * 2;
    auto kern = g3x_kernel<CB>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(p.mbn * p.cbn, p.ksplit), dim3(256), lds, s, p);
    VS_CHECK_LAUNCH();
    return reduce_slabs > 0 ? g3_reduce_launch<CB, G3_K3>(reduce_ws, dw, m_real, c_real, p.mbn, p.cbn, reduce_slabs, s) : VS_OK;
}

// One layer's kernel into `workspace`.  prior_slabs >= 0: reduce (prior_slabs + this layer's) slabs starting prior_slabs slabs BEFORE
// `workspace` into dw; prior_slabs < 0: no reduction (an earlier part of a multi-use weight).  *slabs_out = this layer's slab count.
static int wgrad_single(const void* P, const double* p_stats, const void* Q, const double* q_stats, float* dw,
                        void* workspace, size_t workspace_bytes, int n, int dp, int hp, int wp, int m_ch, int c_ch,
                        int m_real, int c_real, int kind, int dtype, float eps, void* stream, int prior_slabs, int* slabs_out);

extern "C" int vs_conv_wgrad(const void* P, const double* p_stats, const void* Q, const double* q_stats, float* dw,
                             void* workspace, size_t workspace_bytes, int n, int dp, int hp, int wp, int m_ch, int c_ch,
                             int m_real, int c_real, int kind, int dtype, float eps, void* stream) {
    return wgrad_single(P, p_stats, Q, q_stats, dw, workspace, workspace_bytes, n, dp, hp, wp, m_ch, c_ch, m_real, c_real, kind, dtype, eps, stream, 0, nullptr);
}

static int wgrad_single(const void* P, const double* p_stats, const void* Q, const double* q_stats, float* dw,
                        void* workspace, size_t workspace_bytes, int n, int dp, int hp, int wp, int m_ch, int c_ch,
                        int m_real, int c_real, int kind, int dtype, float eps, void* stream, int prior_slabs, int* slabs_out) {
    if (!P || !Q || !dw || !workspace) return VS_EINVAL;
    if (n <= 0 || n > G3_MAXN || dp <= 0 || hp <= 0 || wp <= 0) return VS_ESHAPE;
    if (m_ch % 8 || c_ch % 8 || m_real > m_ch || c_real > c_ch || m_real <= 0 || c_real <= 0) return VS_ESHAPE;
    if (kind != VS_CONV_K3 && kind != VS_CONV_K2S2) return VS_EINVAL;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    G3Params p{};
    int cbsz, ncb;
    g3_plan(n, dp, hp, wp, m_ch, c_ch, kind, cbsz, p.mbn, p.cbn, ncb, p.tiles_per_sample, p.tyn, p.txn, p.ksplit, dtype == VS_F32);
    const size_t need = (size_t)p.ksplit * p.mbn * p.cbn * ncb * 256 * 4;
    if (workspace_bytes < need) return VS_EWORKSPACE;
    p.P = P; p.P_stats = p_stats; p.Q = Q; p.Q_stats = q_stats; p.ws = (float*)workspace;
    p.N = n; p.Dp = dp; p.Hp = hp; p.Wp = wp;
    const int s = kind == VS_CONV_K3 ? 1 : 2;
    p.Dq = dp * s; p.Hq = hp * s; p.Wq = wp * s;
    p.Mch = m_ch; p.Cch = c_ch;
    p.total_tiles = p.tiles_per_sample * n;
    g3_fastdiv(p);
    p.eps = eps;
    p.inv_cnt_p = 1.0 / ((double)dp * hp * wp);
    p.inv_cnt_q = 1.0 / ((double)p.Dq * p.Hq * p.Wq);
    hipStream_t st = (hipStream_t)stream;
    if (slabs_out) *slabs_out = p.ksplit;
    const size_t slab_floats = (size_t)p.mbn * p.cbn * ncb * 256;
    const float* rws = prior_slabs >= 0 ? (const float*)workspace - (size_t)prior_slabs * slab_floats : nullptr;
    const int rsl = prior_slabs >= 0 ? prior_slabs + p.ksplit : 0;
#define G3_GO(T) \
    if (kind == VS_CONV_K3) return cbsz == 16 ? g3_run<T, 16, G3_K3>(p, dw, m_real, c_real, st, rws, rsl) : g3_run<T, 8, G3_K3>(p, dw, m_real, c_real, st, rws, rsl); \
    return cbsz == 16 ? g3_run<T, 16, G3_K2S2>(p, dw, m_real, c_real, st, rws, rsl) : g3_run<T, 8, G3_K2S2>(p, dw, m_real, c_real, st, rws, rsl);
    // both kernels address P and Q with signed 32-bit byte offsets (bounds-checked buffer loads)
    if (dtype == VS_F32 && ((long long)n * dp * hp * wp * m_ch * 4 >= 2147483648ll || (long long)n * p.Dq * p.Hq * p.Wq * c_ch * 4 >= 2147483648ll)) return VS_ESHAPE;
    if (dtype == VS_F32) {
        // 3x3x3 layers: limb arithmetic on the bf16 matrix cores (g3x_kernel); VS_F32_LIMBS=0 and the stride-2 kinds: exact-f32 MFMA (g3_kernel)
        const int limbs = vs_cfg().f32_limbs;
        if (limbs && kind == VS_CONV_K3) return cbsz == 16 ? g3x_run<16>(p, dw, m_real, c_real, st, rws, rsl) : g3x_run<8>(p, dw, m_real, c_real, st, rws, rsl);
        G3_GO(float)
    }
#undef G3_GO
    // g3b_kernel addresses P and Q with signed 32-bit byte offsets
    if ((long long)n * dp * hp * wp * m_ch * 2 >= 2147483648ll || (long long)n * p.Dq * p.Hq * p.Wq * c_ch * 2 >= 2147483648ll) return VS_ESHAPE;
#define G3B_GO(T) \
    if (kind == VS_CONV_K3) return cbsz == 16 ? g3b_run<T, 16, G3_K3>(p, dw, m_real, c_real, st, rws, rsl) : g3b_run<T, 8, G3_K3>(p, dw, m_real, c_real, st, rws, rsl); \
    return cbsz == 16 ? g3b_run<T, 16, G3_K2S2>(p, dw, m_real, c_real, st, rws, rsl) : g3b_run<T, 8, G3_K2S2>(p, dw, m_real, c_real, st, rws, rsl);
    if (dtype == VS_BF16) { G3B_GO(unsigned short) }
    G3B_GO(vs_half)
#undef G3B_GO
}


// ---------------------------------------------------------------------------------------------------
// Grouped path (vs_conv_wgrad_multi): see include/vaeseg.h.
// ---------------------------------------------------------------------------------------------------
#include <algorithm>
#include <vector>

// One reduce launch for every layer: weight slabs (kind 0, runtime geometry of g3_reduce_kernel) and bias partials (kind 1).
struct G3RedDesc {
    const float* ws; float* dw;
    int m_real, c_real, mbn, cbn, nslabs, cb, ntaps, ncb;
    int kind, parts;         // parts: slab partitions summed in parallel (power of two <= G3_RED_ROWS); a block covers 256 * G3_RED_ROWS / parts elements
    int swap;                // the launch computed the TRANSPOSED gradient (operands exchanged, multi_plan): slab (m', c', tap') is dw[c'][m'][ntaps - 1 - tap']
};
#define G3_RED_MAX 56
struct G3RedGroup {
    G3RedDesc d[G3_RED_MAX];
    int blk_start[G3_RED_MAX + 1];
    int n;
};

// RR rows of 64 lanes per block; a row = (element group, slab partition).  256-thread blocks: the 10k blocks of this launch were bound by the
// rate at which 1024-thread workgroups can be started, not by their 40 MB of slabs.
#define G3_RED_ROWS 4
__global__ __launch_bounds__(64 * G3_RED_ROWS) void g3_reduce_group_kernel(const G3RedGroup grp) {
    __shared__ double red[G3_RED_ROWS][64];
    __shared__ double red4[4][G3_RED_ROWS][64];
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_RED_MAX; ++i) l += (i < grp.n && b >= grp.blk_start[i]) ? 1 : 0;
    const G3RedDesc d = grp.d[l];
    const int lb = b - grp.blk_start[l];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (d.kind == 1) {                            // bias partials: double [nslabs][c_real]
        const double* src = (const double*)d.ws;
        const int c = lb * 64 + lane;
        double s = 0.0;
        if (c < d.c_real) {
            int sl = part;
            for (; sl + 7 * G3_RED_ROWS < d.nslabs; sl += 8 * G3_RED_ROWS) {      // 8 loads in flight
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(sl + G3_RED_ROWS * j) * d.c_real + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            for (; sl < d.nslabs; sl += G3_RED_ROWS) s += src[(size_t)sl * d.c_real + c];
        }
        red[part][lane] = s;
        __syncthreads();
        if (part == 0 && c < d.c_real) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < G3_RED_ROWS; ++q) tot += red[q][lane];
            d.dw[c] = (float)tot;
        }
        return;
    }
    // weight slabs: the rows of the block are (element group, slab partition) pairs — a layer with one slab (deep layers
    // of a grouped launch) spends no threads on partitions it does not have
    if (d.cb & 0x200) {
        // slabs of a backward-data launch with the fused weight gradient (igemm_k3tw.h): [tap 27][c 8][m 8] floats, one per workgroup of that launch — up to 512 of
        // them, more than any layer of the grouped launches has.  A block takes 16 vectors of 4 elements and splits the slabs 16 ways (4 per wave), so that
        // no thread walks more than 32 slabs and the entry does not outlast the rest of the launch.
        const int el = lane & 15, sp2 = lane >> 4, P = part * 4 + sp2;
        const size_t e = ((size_t)lb * 16 + el) * 4;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        if (e < 1728) {
            const float* src = d.ws + e;
            int sl = P;
            for (; sl + 7 * 16 < d.nslabs; sl += 8 * 16) {
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(src + (size_t)(sl + 16 * j) * 1728);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s4[0] += (double)v[j][0]; s4[1] += (double)v[j][1]; s4[2] += (double)v[j][2]; s4[3] += (double)v[j][3]; }
            }
            for (; sl < d.nslabs; sl += 16) {
                const f32x4 v = *(const f32x4*)(src + (size_t)sl * 1728);
                s4[0] += (double)v[0]; s4[1] += (double)v[1]; s4[2] += (double)v[2]; s4[3] += (double)v[3];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red4[i][part][lane] = s4[i];
        __syncthreads();
        if (P == 0 && e < 1728) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double tot = 0.0;
                for (int q = 0; q < 16; ++q) tot += red4[i][q >> 2][(q & 3) * 16 + el];       // fixed order: bitwise reproducible
                // S[o][c][m] = sum_v Q(v)[c] P(v + o)[m] = dW[m][c][-o]
                const size_t ee = e + i;
                const int mg = (int)(ee & 7), ci = (int)((ee >> 3) & 7), tp = (int)(ee >> 6);
                if (mg < d.m_real && ci < d.c_real) d.dw[((size_t)mg * d.c_real + ci) * 27 + (26 - tp)] = (float)tot;
            }
        }
        return;
    }
    const int parts = d.parts, egrp = part / parts, sp = part - egrp * parts;
    const size_t slab_elems = (size_t)d.mbn * d.cbn * d.ncb * 256;
    // four consecutive slab elements per lane (= rows 4q..4q+3 of one (k, col)): 1 KiB per wave request instead of 256 B
    const size_t e = (((size_t)lb * (G3_RED_ROWS / parts) + egrp) * 64 + lane) * 4;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    if (e < slab_elems) {
        const float* src = d.ws + e;
        int sl = sp;
        for (; sl + 7 * parts < d.nslabs; sl += 8 * parts) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(src + (size_t)(sl + parts * j) * slab_elems);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s4[0] += (double)v[j][0]; s4[1] += (double)v[j][1]; s4[2] += (double)v[j][2]; s4[3] += (double)v[j][3]; }
        }
        for (; sl < d.nslabs; sl += parts) {
            const f32x4 v = *(const f32x4*)(src + (size_t)sl * slab_elems);
            s4[0] += (double)v[0]; s4[1] += (double)v[1]; s4[2] += (double)v[2]; s4[3] += (double)v[3];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red4[i][part][lane] = s4[i];
    __syncthreads();
    if (sp == 0 && e < slab_elems) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double tot = 0.0;
            for (int q = 0; q < parts; ++q) tot += red4[i][egrp * parts + q][lane];
            const size_t ee = e + i;
            const int row = (int)(ee & 15), col = (int)((ee >> 4) & 15);
            const int k = (int)((ee >> 8) % d.ncb);
            const int pair = (int)(ee / ((size_t)d.ncb * 256));
            const int mb = pair / d.cbn, cb = pair - mb * d.cbn;
            int m = mb * 16 + row;
            int c, tap;
            if (d.cb == 16) { c = cb * 16 + col; tap = k; }
            else if (d.cb & 0x100) {                       // M-packed forms (g3b_body): rows (s, co); 8-channel blocks: columns (dx select, ci), block k = (dz, dy);
                const int sx = row >> 3;                  //                                         16-channel blocks: columns ci, block k = (dz, dy, dx select)
                int dzdy, dxs;
                m = row & 7;
                if ((d.cb & 0xff) == 16) { c = cb * 16 + col; dzdy = k >> 1; dxs = k & 1; }
                else { c = col & 7; dzdy = k; dxs = col >> 3; }
                tap = (sx & dxs) ? d.ntaps : 3 * dzdy + 1 + dxs - sx;    // (s = 1, dx = +1) repeats the centre tap: dropped
            }
            else { c = cb * 8 + (col & 7); tap = 2 * k + (col >> 3); }
            if (m < d.m_real && c < d.c_real && tap < d.ntaps) {
                // swap: rows are the conv's INPUT channels and the tap is mirrored (sum_v Q(v) P(v + o) = dW[-o]); m_real / c_real are the launch's
                if (d.swap) d.dw[((size_t)c * d.m_real + m) * d.ntaps + (d.ntaps - 1 - tap)] = (float)tot;
                else d.dw[((size_t)m * d.c_real + c) * d.ntaps + tap] = (float)tot;
            }
        }
    }
}

// Bias-gradient partials of several layers in one launch (bf16 rows of c_ch channels): block lb of a layer sums rows
// lb*rpi + fy + i*nblk*rpi and writes one double per channel; the grouped reduce adds the blocks in a fixed order.
struct G3BiasDesc { const void* g; double* part; long long rows; int c_ch, c_real, nblk, pad_; };
#define G3_BIAS_MAX 16
struct G3BiasGroup {
    G3BiasDesc d[G3_BIAS_MAX];
    int blk_start[G3_BIAS_MAX + 1];
    int n;
};

template <typename T>
__device__ __forceinline__ void bias_partial_body(const G3BiasDesc& d, const int lb, float* s_red) {
    const int tid = threadIdx.x;
    constexpr int EPL = ET<T>::EPL;               // channels per 16-byte fragment: 8 (16-bit storage) or 4 (fp32)
    const int frags = d.c_ch / EPL, fx = tid % frags, fy = tid / frags, rpi = 256 / frags;
    float part[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) part[j] = 0.f;
    // 8 rows in flight per thread (one load per iteration made the ~32 rounds of a 96^3 layer a dependent chain: 26 us for 100 MB)
    const long long vstep = (long long)d.nblk * rpi;
    for (long long v = (long long)lb * rpi + fy; v < d.rows; v += 8 * vstep) {
        u32x4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long vv = v + u * vstep;
            q[u] = vv < d.rows ? *(const u32x4*)((const T*)d.g + vv * d.c_ch + fx * EPL) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float f[EPL];
            frag_unpack(q[u], f, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) part[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) s_red[tid * 8 + j] = part[j];
    __syncthreads();
    for (int ch = tid; ch < d.c_real; ch += 256) {
        const int cx = ch / EPL, j = ch % EPL;
        double tot = 0.0;
        for (int y = 0; y < rpi; ++y) tot += (double)s_red[(y * frags + cx) * 8 + j];
        d.part[(size_t)lb * d.c_real + ch] = tot;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bias_partial_group_kernel(const G3BiasGroup grp) {
    __shared__ float s_red[256 * 8];
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_BIAS_MAX; ++i) l += (i < grp.n && b >= grp.blk_start[i]) ? 1 : 0;
    bias_partial_body<T>(grp.d[l], b - grp.blk_start[l], s_red);
}

// Every bucket of a pass in ONE grid: a workgroup looks up its layer, then runs that layer's instantiation of g3b_body.  Per-bucket launches each pay
// their own ramp and drain — every workgroup's prologue (statistics tables, first loads) and epilogue (slab reduction through LDS, slab store), about
// 20 us of a 93 us launch (profiles/r04_wgrad_ablation.txt), run at the same moment on every CU; in one grid of several workgroups per slot they overlap
// other workgroups' tile loops.  Registers and LDS are those of the largest instantiation (all of the big ones sit at two workgroups per CU anyway).
// The bias gradients' partial sums ride in the same grid as one more variant: their entries re-use G3Params (P = the gradient rows, ws = the partial sums,
// total_tiles = rows, Mch / Cch = stored / real channels, ksplit = blocks), so the separate 22 us launch disappears into the grid's tail.
enum { G3V_K3_16 = 0, G3V_K3_8, G3V_K2S2_16, G3V_K2S2_8, G3V_UP_16, G3V_K3_16_MP, G3V_K3_8_MP, G3V_BIAS, G3V_BIG_MP8, G3V_BIG_A16H8, G3V_COUNT };
#define G3C_TZ 8
#define G3C_TY 8
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void g3b_uber_kernel(const G3Group grp) {      // (left alone the union of the variants' registers is 284)
    const int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < G3_GROUP_MAX; ++i) l += (i < grp.n && b >= grp.wg_start[i]) ? 1 : 0;
    const G3Params p = grp.p[l];
    const int local = b - grp.wg_start[l];
    const int pairs = p.mbn * p.cbn;
    const int rank = (grp.xcd && p.variant != G3V_BIAS && p.ksplit * pairs >= 16) ? g3_xcd_rank(grp.wg_start[l], local, p.ksplit * pairs, grp.xcd > 1 || pairs == 1) : local;
    const int ks = rank / pairs;
    const int bx = rank - ks * pairs;
    if (p.variant == G3V_BIAS) {
        extern __shared__ __attribute__((aligned(16))) char smem_b[];
        const G3BiasDesc d{p.P, (double*)p.ws, (long long)p.total_tiles, p.Mch, p.Cch, p.ksplit, 0};
        bias_partial_body<T>(d, local, (float*)smem_b);
        return;
    }
    switch (p.variant) {                                   // uniform per workgroup
        case G3V_K3_16:    g3b_body<T, 16, G3_K3, false>(p, bx, ks); break;
        case G3V_K3_8:     g3b_body<T, 8, G3_K3, false>(p, bx, ks); break;
        case G3V_K2S2_16:  g3b_body<T, 16, G3_K2S2, false>(p, bx, ks); break;
        case G3V_K2S2_8:   g3b_body<T, 8, G3_K2S2, false>(p, bx, ks); break;
        case G3V_UP_16:    g3b_body<T, 16, G3_UP, false>(p, bx, ks); break;
        case G3V_K3_16_MP: g3b_body<T, 16, G3_K3, true>(p, bx, ks); break;
        case G3V_BIG_MP8:  g3c_body<T, 8, true, G3C_TZ, G3C_TY>(p, bx, ks); break;
        case G3V_BIG_A16H8: g3c_body<T, 8, false, G3C_TZ, G3C_TY>(p, bx, ks); break;
        default:           g3b_body<T, 8, G3_K3, true>(p, bx, ks); break;
    }
}


namespace {
struct MultiLayer {
    G3Params p;
    int cbsz, ncb, kind, m_real, c_real;
    int big;                 // 0, or the G3V_BIG_* variant this layer runs (g3c_body: 8 x 8 x 16 tiles)
    int swap;                // operands exchanged (see multi_plan): the slabs hold dW transposed and tap-mirrored, the reduction writes it back
    int primary;             // index of the first descriptor with the same dw (a weight used several times in one backward pass:
                             // all uses write slabs into one contiguous region and ONE reduction sums them); == own index otherwise
    int total_slabs;         // primary only: slabs of all its parts
    int bias_primary, bias_total_blk;
    int bias_fold;           // the bias partial sums come out of this layer's own launch (g3b_body, stride-2 kind: bias_g IS its Q operand) — no G3V_BIAS entry
    float* dw;
    size_t ws_off;           // byte offset of this layer's slabs
    long long work;          // tiles per workgroup (sort key)
    // bias
    int bias_nblk;
    size_t bias_off;
};
struct MultiPlan {
    std::vector<MultiLayer> layers;
    size_t bytes = 0;
};

static int multi_validate(const vs_wgrad_desc& d) {
    if (!d.p || !d.q || !d.dw) return VS_EINVAL;
    if (d.n <= 0 || d.n > G3_MAXN || d.dp <= 0 || d.hp <= 0 || d.wp <= 0) return VS_ESHAPE;
    if (d.m_ch % 8 || d.c_ch % 8 || d.m_real > d.m_ch || d.c_real > d.c_ch || d.m_real <= 0 || d.c_real <= 0) return VS_ESHAPE;
    if (d.kind != VS_CONV_K3 && d.kind != VS_CONV_K2S2 && d.kind != VS_CONV_UP) return VS_EINVAL;
    if (d.kind == VS_CONV_UP && (d.reserved_ <= 0 || d.m_ch != 8 * d.reserved_ || d.c_ch < 16 || d.p_stats)) return VS_ESHAPE;     // reserved_ = Co; 16-channel Q blocks only
    if (d.bias_g) {
        if (!d.db || d.bias_rows <= 0 || d.bias_c_real <= 0 || d.bias_c_real > d.bias_c_ch) return VS_EINVAL;
        if (d.bias_c_ch <= 0 || d.bias_c_ch % 8 || d.bias_c_ch > 2048 || 256 % (d.bias_c_ch / 8)) return VS_ESHAPE;
    }
    return VS_OK;
}

// bf16 plan: every layer gets its own slab region; k-splits are chosen per (CB, KIND) bucket so that the bucket's ONE grid has
// about `target` workgroups of about equal tile counts.
static int multi_plan(const vs_wgrad_desc* descs, int count, float eps, MultiPlan& plan, int target_wgs = 0, bool pack_m = false) {
    // workgroups per bucket.  With one grid PER bucket 512 measured best of 384..2560 (two workgroups per CU are resident); inside the all-buckets grid the
    // buckets share the chip, every workgroup pays a prologue (statistics tables, first loads) and an epilogue (slab reduction through LDS, slab store, its
    // share of the reduction launch), and fewer, longer workgroups win: same-box A/B at 96^3 (profiles/r05_ab_wgrad_group_wgs.json)
    // 768 / 512 / 320 / 256 / 160 -> 2.514 / 2.501 / 2.497 / 2.480 / 2.493 ms per step
    // ... and at 160^3 512 / 384 / 256 -> 6.373 / 6.380 / 6.396, at 128^3 (B = 1) 512 / 256 -> 3.737 / 3.729: the small target where the largest layer is small
    const long long target_env = vs_cfg().wgrad_group_wgs;
    long long max_voxels = 0;
    for (int i = 0; i < count; ++i) max_voxels = std::max(max_voxels, (long long)descs[i].n * descs[i].dp * descs[i].hp * descs[i].wp);
    const long long target_default = target_env > 0 ? target_env : (max_voxels <= 2500000 ? 256 : 512);
    const long long target = target_wgs > 0 ? target_wgs : target_default;
    plan.layers.resize(count);
    long long bucket_work[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // (cbsz 16 | 8) x (K3, K2S2, UP, K3 M-packed): one grouped launch each; 8, 9: the big-tile variants
    auto bucket_of = [](const MultiLayer& L) {
        if (L.big) return L.big == G3V_BIG_MP8 ? 8 : 9;
        return (L.cbsz == 16 ? 0 : 1) + 2 * (L.p.mp ? 3 : (L.kind == VS_CONV_K3 ? 0 : (L.kind == VS_CONV_K2S2 ? 1 : 2)));
    };
    // big tiles (g3c_body) for the 3x3x3 layers whose halo operand has 8 stored channels and whose tensors are large enough for the halo traffic to matter;
    // only inside the all-buckets grid (the per-bucket launches of VS_WGRAD_UBER=0 keep the 4x4x16 kernels).  VS_WGRAD_BIG=0 switches it off.
    const bool big_on = pack_m && vs_cfg().wgrad_big != 0 && vs_cfg().wgrad_uber != 0;
    const long long big_min_voxels = vs_cfg().wgrad_big_min_voxels;      // read per plan: the tests lower it (vs_set_config)
    // Operand exchange (round 5).  A 3x3x3 layer's gradient dW[m][c][o] = sum_v P(v)[m] Q(v + o)[c] stages Q with a halo (2.5 x the tile) and P without;
    // when Q is a LAZY activation (statistics given) every halo fragment is normalised + ReLU'd + masked on its way to LDS — 24-28 vector instructions per
    // 16-byte fragment in a kernel that is bound by instruction issue (profiles/r04_wgrad_counters_raw.txt).  The same sums with the roles exchanged,
    // dW'[c][m][o'] = sum_v Q(v)[c] P(v + o')[m] = dW[m][c][-o'], put the halo on P — a stored gradient, staged as it is, out-of-volume fragments already
    // zero from the bounds-checked load — and normalise Q once per voxel.  The reduction writes the transposed, tap-mirrored slabs back (G3RedDesc.swap).
    // Taken when it does not widen the halo operand (m_ch <= c_ch); VS_WGRAD_SWAP=0 switches it off.
    const bool swap_on = pack_m && vs_cfg().wgrad_swap != 0;        // read per plan: the tests and the A/B runs switch it between calls (vs_set_config)
    const bool fold_on = pack_m && vs_cfg().wgrad_uber != 0 && vs_cfg().wgrad_bias_fold != 0;
    std::vector<vs_wgrad_desc> eff(descs, descs + count);
    for (int i = 0; i < count; ++i) {
        int rc = multi_validate(descs[i]);
        if (rc) return rc;
        vs_wgrad_desc& d = eff[i];
        MultiLayer& L = plan.layers[i];
        L.swap = (d.kind == VS_CONV_K3 && d.reserved_ == -1) ? 1 : 0;        // exchanged by the caller already (fp32 path: f32_effective)
        if (!L.swap && swap_on && d.kind == VS_CONV_K3 && d.q_stats != nullptr && d.p_stats == nullptr && d.m_ch <= d.c_ch) {
            std::swap(d.p, d.q); std::swap(d.p_stats, d.q_stats); std::swap(d.m_ch, d.c_ch); std::swap(d.m_real, d.c_real);
            L.swap = 1;
        }
        G3Params& p = L.p;
        p = G3Params{};
        int ks_unused;
        g3_plan(d.n, d.dp, d.hp, d.wp, d.m_ch, d.c_ch, d.kind, L.cbsz, p.mbn, p.cbn, L.ncb, p.tiles_per_sample, p.tyn, p.txn, ks_unused, false);
        p.P = d.p; p.P_stats = d.p_stats; p.Q = d.q; p.Q_stats = d.q_stats;
        p.N = d.n; p.Dp = d.dp; p.Hp = d.hp; p.Wp = d.wp;
        const int s = d.kind != VS_CONV_K2S2 ? 1 : 2;
        p.Dq = d.dp * s; p.Hq = d.hp * s; p.Wq = d.wp * s;
        p.Mch = d.m_ch; p.Cch = d.c_ch;
        p.up_co = d.kind == VS_CONV_UP ? d.reserved_ : 0;
#ifdef VS_G3B_ABLATE
        if (d.kind == VS_CONV_K3 && getenv("VS_G3B_ABLATE")) p.up_co = atoi(getenv("VS_G3B_ABLATE"));
#endif
        p.total_tiles = p.tiles_per_sample * d.n;
        g3_fastdiv(p);
        // g3b_body's M-packed forms.  Same-box A/B (profiles/r04_ab_wgrad_mpack.json): with one grid per bucket 160^3 B=2 6.74 -> 6.50 ms, 128^3 B=1 3.80 -> 3.75, but
        // 96^3 B=2 2.515 -> 2.526 (the packed layers were two more launches); inside the all-buckets grid (g3b_uber_kernel) 96^3 gains too: 2.531 -> 2.513.
        // VS_WGRAD_MPACK=0 switches it off.
        const bool mp_on = vs_cfg().wgrad_mpack != 0;            // read per plan: the tests switch it between calls (vs_set_config)
        if (pack_m && mp_on && d.kind == VS_CONV_K3 && d.m_ch == 8 && (L.cbsz == 16 || d.c_ch == 8)) { p.mp = 1; L.ncb = L.cbsz == 16 ? 18 : 9; }
        L.big = 0;
        if (big_on && d.kind == VS_CONV_K3 && d.c_ch == 8 && (long long)d.n * d.dp * d.hp * d.wp >= big_min_voxels) {
            if (p.mp) L.big = G3V_BIG_MP8;
            else if (d.m_ch % 16 == 0) L.big = G3V_BIG_A16H8;
            if (L.big) {                                   // re-tile: G3C_TZ x G3C_TY x 16
                p.tyn = (d.hp + G3C_TY - 1) / G3C_TY;
                p.tiles_per_sample = ((d.dp + G3C_TZ - 1) / G3C_TZ) * p.tyn * p.txn;
                p.total_tiles = p.tiles_per_sample * d.n;
                g3_fastdiv(p);
            }
        }
        p.eps = eps;
        p.inv_cnt_p = 1.0 / ((double)d.dp * d.hp * d.wp);
        p.inv_cnt_q = 1.0 / ((double)p.Dq * p.Hq * p.Wq);
        if ((long long)d.n * d.dp * d.hp * d.wp * d.m_ch * 2 >= 2147483648ll || (long long)d.n * p.Dq * p.Hq * p.Wq * d.c_ch * 2 >= 2147483648ll) return VS_ESHAPE;
        L.kind = d.kind; L.m_real = d.m_real; L.c_real = d.c_real; L.dw = d.dw;
        bucket_work[bucket_of(L)] += (long long)p.mbn * p.cbn * p.total_tiles;
    }
    // descriptors that share dw (db): parts of one gradient
    for (int i = 0; i < count; ++i) {
        MultiLayer& L = plan.layers[i];
        L.primary = i; L.bias_primary = i;
        for (int j = 0; j < i; ++j) {
            if (descs[j].dw == descs[i].dw && L.primary == i) {
                const MultiLayer& F = plan.layers[j];
                if (F.cbsz != L.cbsz || F.kind != L.kind || F.p.mbn != L.p.mbn || F.p.cbn != L.p.cbn || F.m_real != L.m_real || F.c_real != L.c_real || F.p.mp != L.p.mp || F.swap != L.swap
                    || (F.big != 0) != (L.big != 0))
                    return VS_EINVAL;                     // same destination, different layer geometry
                L.primary = F.primary;
            }
            if (descs[i].bias_g && descs[j].bias_g && descs[j].db == descs[i].db && L.bias_primary == i) {
                if (descs[j].bias_c_real != descs[i].bias_c_real) return VS_EINVAL;
                L.bias_primary = plan.layers[j].bias_primary;
            }
        }
    }
    size_t off = 0;
    for (int i = 0; i < count; ++i) {                      // k-splits
        MultiLayer& L = plan.layers[i];
        G3Params& p = L.p;
        const long long w = bucket_work[bucket_of(L)];
        long long tpw = (w + target - 1) / target;            // tiles per workgroup
        if (tpw < 1) tpw = 1;
        long long ks = (p.total_tiles + tpw - 1) / tpw;
        const double slab_bytes = (double)p.mbn * p.cbn * L.ncb * 256 * 4;
        while (ks > 1 && ks * slab_bytes > 64.0 * 1024 * 1024) --ks;
        p.ksplit = (int)ks;
        L.work = (p.total_tiles + ks - 1) / ks;
        L.total_slabs = 0; L.bias_total_blk = 0;
        const vs_wgrad_desc& d = descs[i];
        L.bias_nblk = 0; L.bias_off = 0;
        L.bias_fold = 0;
        if (d.bias_g) {
            const int rpi = 256 / (d.bias_c_ch / 8);
            long long nb = (d.bias_rows + (long long)rpi * 32 - 1) / ((long long)rpi * 32);
            L.bias_nblk = (int)std::min<long long>(std::max<long long>(nb, 1), target_wgs > 0 ? std::max(8, target_wgs / 2) : 256);     // <= 2 rounds of the reduce's 16 x 8 loads
            // ConvTranspose3d's bias: the gradient rows are the fine tensor the stride-2 kernel stages as Q — summed there (all-buckets grid, 16-bit storage)
            if (fold_on && d.kind == VS_CONV_K2S2 && d.bias_g == d.q && d.bias_c_ch == d.c_ch && d.bias_c_real == d.c_ch &&
                d.bias_rows == (long long)d.n * p.Dq * p.Hq * p.Wq) {
                L.bias_fold = 1;
                L.bias_nblk = p.ksplit;
            }
        }
    }
    for (int i = 0; i < count; ++i) {                      // slab regions: the parts of one gradient lie back to back
        if (plan.layers[i].primary != i) continue;
        for (int j = i; j < count; ++j) {
            MultiLayer& L = plan.layers[j];
            if (L.primary != i) continue;
            L.ws_off = off;
            off += (size_t)L.p.ksplit * L.p.mbn * L.p.cbn * L.ncb * 256 * 4;
            plan.layers[i].total_slabs += L.p.ksplit;
        }
    }
    for (int i = 0; i < count; ++i) {                      // bias partial regions, likewise
        if (!descs[i].bias_g || plan.layers[i].bias_primary != i) continue;
        for (int j = i; j < count; ++j) {
            MultiLayer& L = plan.layers[j];
            if (!descs[j].bias_g || L.bias_primary != i) continue;
            L.bias_off = off;
            off += (size_t)L.bias_nblk * descs[j].bias_c_real * sizeof(double);
            plan.layers[i].bias_total_blk += L.bias_nblk;
            if (L.bias_fold) {
                const size_t rel = (L.bias_off - L.ws_off) / sizeof(double);          // bias regions lie behind every slab region; both are multiples of 8 bytes
                if ((L.bias_off - L.ws_off) % sizeof(double) || rel == 0 || rel >= 2147483647ull) return VS_ESHAPE;
                L.p.up_co = (int)rel;
            }
        }
        off = (off + 255) / 256 * 256;
    }
    plan.bytes = off;
    return VS_OK;
}

template <typename T, int CB, int KIND, bool MP = false>
static int g3b_group_run(const G3Group& grp, hipStream_t s) {
    using GEO = G3Geo<CB, KIND>;
    constexpr size_t lds = G3B_LDS_Q + (size_t)GEO::QV * CB * 2;
    auto kern = g3b_group_kernel<CB, KIND, T, MP>;
    if (lds > 64 * 1024) {
        static const hipError_t attr_err =
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr_err != hipSuccess) return (int)attr_err;
    }
    hipLaunchKernelGGL(kern, dim3(grp.wg_start[grp.n]), dim3(256), lds, s, grp);
    VS_CHECK_LAUNCH();
    return VS_OK;
}


static int g3v_of(int cbsz, int kind, bool mp, int big = 0) {
    if (big) return big;
    if (kind == VS_CONV_UP) return G3V_UP_16;
    if (kind == VS_CONV_K2S2) return cbsz == 16 ? G3V_K2S2_16 : G3V_K2S2_8;
    if (mp) return cbsz == 16 ? G3V_K3_16_MP : G3V_K3_8_MP;
    return cbsz == 16 ? G3V_K3_16 : G3V_K3_8;
}
static size_t g3v_lds(int variant) {
    switch (variant) {
        case G3V_K3_16: case G3V_K3_16_MP: return G3B_LDS_Q + (size_t)G3Geo<16, G3_K3>::QV * 16 * 2;
        case G3V_K3_8: case G3V_K3_8_MP:   return G3B_LDS_Q + (size_t)G3Geo<8, G3_K3>::QV * 8 * 2;
        case G3V_K2S2_16:                  return G3B_LDS_Q + (size_t)G3Geo<16, G3_K2S2>::QV * 16 * 2;
        case G3V_K2S2_8:                   return G3B_LDS_Q + (size_t)G3Geo<8, G3_K2S2>::QV * 8 * 2;
        case G3V_BIG_MP8:                  return G3CGeo<8, true, G3C_TZ, G3C_TY>::LDS;
        case G3V_BIG_A16H8:                return G3CGeo<8, false, G3C_TZ, G3C_TY>::LDS;
        default:                           return G3B_LDS_Q + (size_t)G3Geo<16, G3_UP>::QV * 16 * 2;
    }
}
// relative duration of one tile of a variant's loop (sort key of the all-buckets launch: longest workgroups first)
static double g3v_tile_cost(int variant) {
    switch (variant) {
        case G3V_K3_16: return 2.0; case G3V_K3_16_MP: return 1.5; case G3V_K3_8: return 1.0; case G3V_K3_8_MP: return 0.9;
        case G3V_K2S2_16: return 1.6; case G3V_K2S2_8: return 0.8; case G3V_BIG_MP8: return 3.2; case G3V_BIG_A16H8: return 4.5; default: return 2.0;
    }
}
template <typename T>
static int g3b_uber_run(const G3Group& grp, size_t lds, hipStream_t s) {
    auto kern = g3b_uber_kernel<T>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    if (lds > 96 * 1024) return VS_ESHAPE;
    hipLaunchKernelGGL(kern, dim3(grp.wg_start[grp.n]), dim3(256), lds, s, grp);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- fp32 parity mode: the 3x3x3 layers of a pass as grouped limb launches (g3x_group_kernel), one grid per channel-block width ----------
static bool f32_limbs_on() {
    const int on = vs_cfg().f32_limbs;
    return on != 0;
}
// descriptors of the pass that take the limb path, bias requests stripped (the fp32 branch sums biases with vs_bias_grad_acc); cb: 16 / 8
// the fp32 mode's grouped launches: 3x3x3 layers on the limb kernels, stride-2 layers on the exact-f32 MFMA body; 16- and 8-channel blocks each
struct F32Grp { int kind, cb; };
static const F32Grp F32_GROUPS[4] = {{VS_CONV_K3, 16}, {VS_CONV_K3, 8}, {VS_CONV_K2S2, 16}, {VS_CONV_K2S2, 8}};
// fp32 parity mode: the operand exchange of multi_plan applied BEFORE the layers are sorted into channel-block groups (the exchange changes which operand's
// width picks the group): a 16 -> 8 layer at full resolution becomes 16 full MFMA rows against 14 two-tap blocks of the 8-channel gradient instead of 8 of 16
// rows against 27 blocks — 84 instead of 162 limb MFMAs per 32 voxels — and the halo operand is the stored gradient (no normalise before the limb split).
// Marked with reserved_ = -1 (a field only VS_CONV_UP descriptors use).  VS_WGRAD_SWAP=0 switches it off.
static std::vector<vs_wgrad_desc> f32_effective(const vs_wgrad_desc* descs, int count) {
    std::vector<vs_wgrad_desc> eff(descs, descs + count);
    if (vs_cfg().wgrad_swap == 0 || !f32_limbs_on()) return eff;      // only the grouped limb launches write the exchanged form back (G3RedDesc.swap)
    for (vs_wgrad_desc& d : eff) {
        if (d.kind == VS_CONV_K3 && d.q_stats != nullptr && d.p_stats == nullptr && d.m_ch <= d.c_ch && d.p && d.q) {
            std::swap(d.p, d.q); std::swap(d.p_stats, d.q_stats); std::swap(d.m_ch, d.c_ch); std::swap(d.m_real, d.c_real);
            d.reserved_ = -1;
        }
    }
    return eff;
}
static bool f32_grouped_kind(int kind) { return kind == VS_CONV_K2S2 || (kind == VS_CONV_K3 && f32_limbs_on()); }
static std::vector<vs_wgrad_desc> f32_limb_subset(const vs_wgrad_desc* descs, int count, int cb, int kind = VS_CONV_K3) {
    std::vector<vs_wgrad_desc> out;
    if (!f32_grouped_kind(kind)) return out;
    for (int i = 0; i < count; ++i) {
        if (descs[i].kind != kind || (descs[i].c_ch >= 16 ? 16 : 8) != cb) continue;
        vs_wgrad_desc d = descs[i];
        d.bias_g = nullptr; d.db = nullptr; d.bias_rows = 0; d.bias_c_ch = 0; d.bias_c_real = 0;
        out.push_back(d);
    }
    return out;
}
static inline int f32_limb_target(int cb) { return cb == 16 ? 256 : 512; }       // resident workgroups: one per CU (16-channel blocks: 90 KB of LDS), two (8)
// stride-2 layers (exact-f32 MFMA, fp32 accumulators): as many workgroups as keep a workgroup at <= 8 tiles — the bound on an accumulator's chain the
// per-layer plan keeps (g3_plan: VS_WGRAD_F32_TILES)
static int f32_group_target(const std::vector<vs_wgrad_desc>& sub, int cb, int kind) {
    if (kind == VS_CONV_K3) return f32_limb_target(cb);
    long long work = 0;
    for (const vs_wgrad_desc& d : sub) {
        int cbsz, mbn, cbn, ncb, tps, tyn, txn, ks;
        g3_plan(d.n, d.dp, d.hp, d.wp, d.m_ch, d.c_ch, d.kind, cbsz, mbn, cbn, ncb, tps, tyn, txn, ks, false);
        work += (long long)mbn * cbn * tps * d.n;
    }
    return (int)std::min<long long>(std::max<long long>(256, (work + 7) / 8), 1 << 20);
}
template <int CB, int KIND>
static int g3_group_run(const G3Group& grp, hipStream_t s) {
    using GEO = G3Geo<CB, KIND>;
    constexpr size_t lds = G3_LDS_Q + (size_t)GEO::QV * CB * 4;
    auto kern = g3_group_kernel<CB, KIND>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(grp.wg_start[grp.n]), dim3(256), lds, s, grp);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <int CB>
static int g3x_group_run(const G3Group& grp, hipStream_t s) {
    using GEO = G3Geo<CB, G3_K3>;
    constexpr size_t lds = G3X_LDS_Q + (size_t)3 * GEO::QV * CB * 2;
    auto kern = g3x_group_kernel<CB>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(grp.wg_start[grp.n]), dim3(256), lds, s, grp);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

static int f32_limb_group_launch(const std::vector<vs_wgrad_desc>& sub, int cb, float eps, char* ws, size_t ws_bytes, hipStream_t st, int kind = VS_CONV_K3) {
    if (sub.empty()) return VS_OK;
    MultiPlan plan;
    int rc = multi_plan(sub.data(), (int)sub.size(), eps, plan, f32_group_target(sub, cb, kind));
    if (rc) return rc;
    if (ws_bytes < plan.bytes) return VS_EWORKSPACE;
    const int count = (int)sub.size();
    for (int i = 0; i < count; ++i) {                    // fp32 operands: 4-byte elements under the 32-bit buffer offsets
        const vs_wgrad_desc& d = sub[i];
        const long long qv = kind == VS_CONV_K2S2 ? 8 : 1;          // Q grid: the fine one for the stride-2 kinds
        if ((long long)d.n * d.dp * d.hp * d.wp * d.m_ch * 4 >= 2147483648ll || (long long)d.n * d.dp * d.hp * d.wp * qv * d.c_ch * 4 >= 2147483648ll) return VS_ESHAPE;
    }
    std::vector<int> idx(count);
    for (int i = 0; i < count; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return plan.layers[a].work > plan.layers[b].work; });
    for (size_t at = 0; at < idx.size(); at += G3_GROUP_MAX) {
        G3Group grp{};
        grp.xcd = vs_cfg().wgrad_xcd;                    // >= 2: g3_xcd_rank (the channel-block pairs of a tile on one XCD)
        grp.n = (int)std::min<size_t>(G3_GROUP_MAX, idx.size() - at);
        long long wg = 0;
        for (int j = 0; j < grp.n; ++j) {
            MultiLayer& L = plan.layers[idx[at + j]];
            grp.p[j] = L.p;
            grp.p[j].ws = (float*)(ws + L.ws_off);
            grp.wg_start[j] = (int)wg;
            wg += (long long)L.p.mbn * L.p.cbn * L.p.ksplit;
        }
        if (wg >= 2147483647ll) return VS_ESHAPE;
        for (int j = grp.n; j <= G3_GROUP_MAX; ++j) grp.wg_start[j] = (int)wg;
        if (kind == VS_CONV_K2S2) rc = cb == 16 ? g3_group_run<16, G3_K2S2>(grp, st) : g3_group_run<8, G3_K2S2>(grp, st);
        else rc = cb == 16 ? g3x_group_run<16>(grp, st) : g3x_group_run<8>(grp, st);
        if (rc) return rc;
    }
    std::vector<G3RedDesc> red;
    std::vector<int> blocks;
    for (int i = 0; i < count; ++i) {
        const MultiLayer& L = plan.layers[i];
        if (L.primary != i) continue;
        const long long slab_elems = (long long)L.p.mbn * L.p.cbn * L.ncb * 256;
        int parts = 1;
        while (parts < G3_RED_ROWS && parts * 8 < L.total_slabs) parts *= 2;
        red.push_back(G3RedDesc{(const float*)(ws + L.ws_off), L.dw, L.m_real, L.c_real, L.p.mbn, L.p.cbn, L.total_slabs, L.cbsz, kind == VS_CONV_K2S2 ? 8 : 27, L.ncb, 0, parts, L.swap});
        blocks.push_back(vs_ceil_div(slab_elems, 256 * (G3_RED_ROWS / parts)));
    }
    for (size_t at = 0; at < red.size(); at += G3_RED_MAX) {
        G3RedGroup grp{};
        grp.n = (int)std::min<size_t>(G3_RED_MAX, red.size() - at);
        long long blk = 0;
        for (int j = 0; j < grp.n; ++j) { grp.d[j] = red[at + j]; grp.blk_start[j] = (int)blk; blk += blocks[at + j]; }
        if (blk >= 2147483647ll) return VS_ESHAPE;
        for (int j = grp.n; j <= G3_RED_MAX; ++j) grp.blk_start[j] = (int)blk;
        hipLaunchKernelGGL(g3_reduce_group_kernel, dim3((unsigned)blk), dim3(64 * G3_RED_ROWS), 0, st, grp);
        VS_CHECK_LAUNCH();
    }
    return VS_OK;
}
static size_t f32_limb_group_bytes(const std::vector<vs_wgrad_desc>& sub, int cb, int kind = VS_CONV_K3) {
    if (sub.empty()) return 0;
    MultiPlan plan;
    if (multi_plan(sub.data(), (int)sub.size(), 0.f, plan, f32_group_target(sub, cb, kind))) return 0;
    return (plan.bytes + 255) / 256 * 256;
}
}  // namespace

// the grouped launches with a caller-chosen grid width (target_workgroups; 0 = the default, 512: two per CU).  Internal since round 5: the
// early, throttled branch it was exported for (round 4) measured slower at every width (profiles/r04_ab_wgrad_early.json).
static size_t wgrad_multi_workspace_bytes_impl(const vs_wgrad_desc* descs, int count, int dtype, int target_workgroups);
static int wgrad_multi_impl(const vs_wgrad_desc* descs, int count, void* workspace, size_t workspace_bytes, int dtype,
                            float eps, int target_workgroups, void* stream);
extern "C" size_t vs_conv_wgrad_multi_workspace_bytes(const vs_wgrad_desc* descs, int count, int dtype) {
    return wgrad_multi_workspace_bytes_impl(descs, count, dtype, 0);
}
extern "C" int vs_conv_wgrad_multi(const vs_wgrad_desc* descs, int count, void* workspace, size_t workspace_bytes, int dtype,
                                   float eps, void* stream) {
    return wgrad_multi_impl(descs, count, workspace, workspace_bytes, dtype, eps, 0, stream);
}

// fp32 mode: the bias gradients of a pass as the 16-bit path computes them (grouped partial sums, one fixed-order reduction) instead of a zero fill and a float-atomic
// launch per layer.  The plan is multi_plan's; its bias regions follow its slab regions, so they are rebased to the start of the first one.
static size_t f32_bias_base(const vs_wgrad_desc* descs, int count, const MultiPlan& plan) {
    size_t base = (size_t)-1;
    for (int i = 0; i < count; ++i) if (descs[i].bias_g) base = std::min(base, plan.layers[i].bias_off);
    return base;
}
static size_t f32_bias_bytes(const vs_wgrad_desc* descs, int count, int target_workgroups) {
    bool any = false;
    for (int i = 0; i < count; ++i) any = any || descs[i].bias_g != nullptr;
    if (!any) return 0;
    MultiPlan plan;
    if (multi_plan(descs, count, 0.f, plan, target_workgroups)) return 0;
    return plan.bytes - f32_bias_base(descs, count, plan);
}

// VS_WGRAD_SLABS descriptors (slabs already computed by vs_conv_k3_bwd_data_wgrad): no grid work, no workspace, one entry of the grouped reduction each
static std::vector<vs_wgrad_desc> without_slab_descs(const vs_wgrad_desc* descs, int count) {
    std::vector<vs_wgrad_desc> reg;
    for (int i = 0; i < count; ++i) if (descs[i].kind != VS_WGRAD_SLABS) reg.push_back(descs[i]);
    return reg;
}
static int slab_desc_validate(const vs_wgrad_desc& d) {
    if (!d.p || !d.dw || d.n <= 0 || ((uintptr_t)d.p & 15)) return VS_EINVAL;
    if (d.bias_g && (!d.db || d.bias_c_real <= 0 || d.bias_c_real > 8 || d.bias_rows != d.n || ((uintptr_t)d.bias_g & 7))) return VS_EINVAL;    // double [n][bias_c_real] partials
    if (d.m_real <= 0 || d.m_real > 8 || d.c_real <= 0 || d.c_real > 8 || d.m_ch != 8 || d.c_ch != 8) return VS_ESHAPE;
    return VS_OK;
}
static void slab_red_entry(const vs_wgrad_desc& d, G3RedDesc& r, int& blocks) {
    r = G3RedDesc{(const float*)d.p, d.dw, d.m_real, d.c_real, 1, 1, d.n, 0x208, 27, 27, 0, G3_RED_ROWS, 0};
    blocks = 1728 / 64;                                  // 16 vectors of 4 elements per block (g3_reduce_group_kernel)
}
// the bias partials a slab descriptor may carry (vs_conv_k3_softmax2_bwd_data): one more entry of the bias kind
static void slab_bias_entry(const vs_wgrad_desc& d, G3RedDesc& r, int& blocks) {
    r = G3RedDesc{(const float*)d.bias_g, d.db, 0, d.bias_c_real, 0, 0, d.n, 0, 0, 0, 1, G3_RED_ROWS, 0};
    blocks = vs_ceil_div(d.bias_c_real, 64);
}
static int slab_reduce_only(const std::vector<vs_wgrad_desc>& slabs, hipStream_t st) {
    std::vector<G3RedDesc> ents;
    std::vector<int> nbs;
    for (const vs_wgrad_desc& sd : slabs) {
        G3RedDesc r;
        int nb = 0;
        slab_red_entry(sd, r, nb);
        ents.push_back(r); nbs.push_back(nb);
        if (sd.bias_g) { slab_bias_entry(sd, r, nb); ents.push_back(r); nbs.push_back(nb); }
    }
    for (size_t at = 0; at < ents.size(); at += G3_RED_MAX) {
        G3RedGroup grp{};
        grp.n = (int)std::min<size_t>(G3_RED_MAX, ents.size() - at);
        long long blk = 0;
        for (int j = 0; j < grp.n; ++j) {
            grp.d[j] = ents[at + j];
            grp.blk_start[j] = (int)blk;
            blk += nbs[at + j];
        }
        for (int j = grp.n; j <= G3_RED_MAX; ++j) grp.blk_start[j] = (int)blk;
        hipLaunchKernelGGL(g3_reduce_group_kernel, dim3((unsigned)blk), dim3(64 * G3_RED_ROWS), 0, st, grp);
        VS_CHECK_LAUNCH();
    }
    return VS_OK;
}

static size_t wgrad_multi_workspace_bytes_impl(const vs_wgrad_desc* descs, int count, int dtype, int target_workgroups) {
    if (!descs || count <= 0 || target_workgroups < 0) return 0;
    std::vector<vs_wgrad_desc> reg_only;
    {
        bool any = false;
        for (int i = 0; i < count; ++i) any = any || descs[i].kind == VS_WGRAD_SLABS;
        if (any) {
            if (dtype == VS_F32) return 0;
            reg_only = without_slab_descs(descs, count);
            if (reg_only.empty()) return 256;               // nothing but reductions: a token workspace (the caller passes a non-null pointer)
            descs = reg_only.data(); count = (int)reg_only.size();
        }
    }
    std::vector<vs_wgrad_desc> eff32;
    if (dtype == VS_F32) {                        // serial per-layer launches share one region; the uses of one weight need theirs side by side
        eff32 = f32_effective(descs, count);
        descs = eff32.data();
        // the 3x3x3 layers of the limb path come first: two grouped regions (16- / 8-channel blocks); the serial region of the other layers follows
        size_t limb_bytes = 0;
        for (const F32Grp& gk : F32_GROUPS) limb_bytes += f32_limb_group_bytes(f32_limb_subset(descs, count, gk.cb, gk.kind), gk.cb, gk.kind);
        size_t mx = 0;
        for (int i = 0; i < count; ++i) {
            if (f32_grouped_kind(descs[i].kind)) continue;
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && descs[j].dw != descs[i].dw;
            if (!first) continue;
            size_t sum = 0;
            for (int j = i; j < count; ++j)
                if (descs[j].dw == descs[i].dw)
                    sum += vs_conv_wgrad_workspace_bytes(descs[j].n, descs[j].dp, descs[j].hp, descs[j].wp, descs[j].m_ch, descs[j].c_ch, descs[j].kind);
            mx = std::max(mx, sum);
        }
        return limb_bytes + (mx + 255) / 256 * 256 + f32_bias_bytes(descs, count, target_workgroups);
    }
    MultiPlan plan;
    if (multi_plan(descs, count, 0.f, plan, target_workgroups, true)) return 0;
    return plan.bytes;
}

static int wgrad_multi_impl(const vs_wgrad_desc* descs, int count, void* workspace, size_t workspace_bytes, int dtype,
                            float eps, int target_workgroups, void* stream) {
    if (!descs || count <= 0 || !workspace || target_workgroups < 0) return VS_EINVAL;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    const bool f16 = dtype == VS_F16;
    hipStream_t st = (hipStream_t)stream;
    std::vector<vs_wgrad_desc> reg_only, slab_descs;
    for (int i = 0; i < count; ++i)
        if (descs[i].kind == VS_WGRAD_SLABS) {
            if (dtype == VS_F32) return VS_EINVAL;
            int rc = slab_desc_validate(descs[i]);
            if (rc) return rc;
            for (int j = 0; j < count; ++j) if (j != i && descs[j].dw == descs[i].dw) return VS_EINVAL;      // a slab descriptor is the only contribution to its gradient
            slab_descs.push_back(descs[i]);
        }
    if (!slab_descs.empty()) {
        reg_only = without_slab_descs(descs, count);
        if (reg_only.empty()) return slab_reduce_only(slab_descs, st);
        descs = reg_only.data(); count = (int)reg_only.size();
    }
    std::vector<vs_wgrad_desc> eff32;
    if (dtype == VS_F32) {
        for (int i = 0; i < count; ++i) {
            int rc = multi_validate(descs[i]);
            if (rc) return rc;
        }
        eff32 = f32_effective(descs, count);
        descs = eff32.data();
        // the 3x3x3 layers: grouped limb launches on the bf16 matrix cores (g3x_group_kernel) — with per-layer launches the 16 small layers of a
        // step cost 30-47 us each (one tile per workgroup under fixed prologue / slab costs); the stride-2 kinds: per-layer exact-f32 kernels
        {
            char* wsp = (char*)workspace;
            size_t left = workspace_bytes;
            for (const F32Grp& gk : F32_GROUPS) {
                const std::vector<vs_wgrad_desc> sub = f32_limb_subset(descs, count, gk.cb, gk.kind);
                const size_t need = f32_limb_group_bytes(sub, gk.cb, gk.kind);
                if (need > left) return VS_EWORKSPACE;
                int rc = f32_limb_group_launch(sub, gk.cb, eps, wsp, left, st, gk.kind);
                if (rc) return rc;
                wsp += need; left -= need;
            }
            workspace = wsp; workspace_bytes = left;
        }
        for (int i = 0; i < count; ++i) {
            if (f32_grouped_kind(descs[i].kind)) continue;
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && descs[j].dw != descs[i].dw;
            if (!first) continue;
            int last = i;
            for (int j = i + 1; j < count; ++j) if (descs[j].dw == descs[i].dw) last = j;
            // the uses of this weight, slabs side by side; the last one's launch reduces them all
            size_t off = 0;
            int prior = 0;
            for (int j = i; j <= last; ++j) {
                const vs_wgrad_desc& d = descs[j];
                if (d.dw != descs[i].dw) continue;
                if (d.m_ch != descs[i].m_ch || d.c_ch != descs[i].c_ch || d.kind != descs[i].kind || d.m_real != descs[i].m_real || d.c_real != descs[i].c_real)
                    return VS_EINVAL;
                int slabs = 0;
                int rc = wgrad_single(d.p, d.p_stats, d.q, d.q_stats, d.dw, (char*)workspace + off, workspace_bytes - off, d.n, d.dp, d.hp, d.wp,
                                      d.m_ch, d.c_ch, d.m_real, d.c_real, d.kind, dtype, eps, stream, j == last ? prior : -1, &slabs);
                if (rc) return rc;
                off += vs_conv_wgrad_workspace_bytes(d.n, d.dp, d.hp, d.wp, d.m_ch, d.c_ch, d.kind) / 1;
                // slabs of the next part start right behind this part's: its plan is the fp32 plan vs_conv_wgrad_workspace_bytes sizes
                prior += slabs;
            }
        }
        bool any_bias = false;
        for (int i = 0; i < count; ++i) any_bias = any_bias || descs[i].bias_g != nullptr;
        if (any_bias) {
            size_t mx = 0;                               // the serial region's size, as the workspace query counts it
            for (int i = 0; i < count; ++i) {
                if (f32_grouped_kind(descs[i].kind)) continue;
                bool first = true;
                for (int j = 0; j < i; ++j) first = first && descs[j].dw != descs[i].dw;
                if (!first) continue;
                size_t sum = 0;
                for (int j = i; j < count; ++j)
                    if (descs[j].dw == descs[i].dw)
                        sum += vs_conv_wgrad_workspace_bytes(descs[j].n, descs[j].dp, descs[j].hp, descs[j].wp, descs[j].m_ch, descs[j].c_ch, descs[j].kind);
                mx = std::max(mx, sum);
            }
            mx = (mx + 255) / 256 * 256;
            MultiPlan plan;
            int rc = multi_plan(descs, count, eps, plan, target_workgroups);
            if (rc) return rc;
            const size_t base = f32_bias_base(descs, count, plan);
            if (workspace_bytes < mx + (plan.bytes - base)) return VS_EWORKSPACE;
            char* bws = (char*)workspace + mx;
            std::vector<int> idx;
            for (int i = 0; i < count; ++i) if (descs[i].bias_g) idx.push_back(i);
            for (size_t at = 0; at < idx.size(); at += G3_BIAS_MAX) {
                G3BiasGroup grp{};
                grp.n = (int)std::min<size_t>(G3_BIAS_MAX, idx.size() - at);
                int blk = 0;
                for (int j = 0; j < grp.n; ++j) {
                    const int i = idx[at + j];
                    const vs_wgrad_desc& d = descs[i];
                    if (d.bias_c_ch % 4 || 256 % (d.bias_c_ch / 4)) return VS_ESHAPE;
                    grp.d[j] = G3BiasDesc{d.bias_g, (double*)(bws + (plan.layers[i].bias_off - base)), d.bias_rows, d.bias_c_ch, d.bias_c_real, plan.layers[i].bias_nblk, 0};
                    grp.blk_start[j] = blk;
                    blk += plan.layers[i].bias_nblk;
                }
                for (int j = grp.n; j <= G3_BIAS_MAX; ++j) grp.blk_start[j] = blk;
                hipLaunchKernelGGL(bias_partial_group_kernel<float>, dim3(blk), dim3(256), 0, st, grp);
                VS_CHECK_LAUNCH();
            }
            std::vector<G3RedDesc> red;
            for (int i = 0; i < count; ++i)
                if (descs[i].bias_g && plan.layers[i].bias_primary == i)
                    red.push_back(G3RedDesc{(const float*)(bws + (plan.layers[i].bias_off - base)), descs[i].db, 0, descs[i].bias_c_real, 0, 0, plan.layers[i].bias_total_blk, 0, 0, 0, 1, G3_RED_ROWS});
            for (size_t at = 0; at < red.size(); at += G3_RED_MAX) {
                G3RedGroup grp{};
                grp.n = (int)std::min<size_t>(G3_RED_MAX, red.size() - at);
                long long blk = 0;
                for (int j = 0; j < grp.n; ++j) { grp.d[j] = red[at + j]; grp.blk_start[j] = (int)blk; blk += vs_ceil_div(red[at + j].c_real, 64); }
                for (int j = grp.n; j <= G3_RED_MAX; ++j) grp.blk_start[j] = (int)blk;
                hipLaunchKernelGGL(g3_reduce_group_kernel, dim3((unsigned)blk), dim3(64 * G3_RED_ROWS), 0, st, grp);
                VS_CHECK_LAUNCH();
            }
        }
        return VS_OK;
    }
    MultiPlan plan;
    int rc = multi_plan(descs, count, eps, plan, target_workgroups, true);
    if (rc) return rc;
    if (workspace_bytes < plan.bytes) return VS_EWORKSPACE;
    if ((uintptr_t)workspace % 16) return VS_EINVAL;
    char* ws = (char*)workspace;

    // ---- every bucket in one grid (G3_GROUP_MAX layers per grid, longest workgroups first); VS_WGRAD_UBER=0: one grid per bucket ----
    const int uber = vs_cfg().wgrad_uber;       // read per call (the tests run both forms)
    if (uber) {
        std::vector<int> idx(count);
        for (int i = 0; i < count; ++i) idx[i] = i;
        auto key = [&](int a) { const MultiLayer& L = plan.layers[a]; return (double)L.work * g3v_tile_cost(g3v_of(L.cbsz, L.kind, L.p.mp != 0, L.big)); };
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) > key(b); });
        for (int i = 0; i < count; ++i)                      // the bias gradients' partial sums: entries -(i + 1), behind the weight layers (short workgroups)
            if (descs[i].bias_g && descs[i].bias_rows < 2147483647ll && !plan.layers[i].bias_fold) idx.push_back(-(i + 1));
        for (size_t at = 0; at < idx.size(); at += G3_GROUP_MAX) {
            G3Group grp{};
            const int xcd_walk = vs_cfg().wgrad_xcd;
            grp.xcd = xcd_walk;
            grp.n = (int)std::min<size_t>(G3_GROUP_MAX, idx.size() - at);
            long long wg = 0;
            size_t lds = 0;
            for (int j = 0; j < grp.n; ++j) {
                if (idx[at + j] < 0) {
                    const int i = -idx[at + j] - 1;
                    const vs_wgrad_desc& d = descs[i];
                    G3Params q{};
                    q.P = d.bias_g; q.ws = (float*)(ws + plan.layers[i].bias_off);
                    q.total_tiles = (int)d.bias_rows; q.Mch = d.bias_c_ch; q.Cch = d.bias_c_real; q.ksplit = plan.layers[i].bias_nblk;
                    q.mbn = 1; q.cbn = 1; q.variant = G3V_BIAS;
                    grp.p[j] = q;
                    lds = std::max(lds, (size_t)256 * 8 * sizeof(float));
                    grp.wg_start[j] = (int)wg;
                    wg += q.ksplit;
                    continue;
                }
                MultiLayer& L = plan.layers[idx[at + j]];
                if (L.kind == VS_CONV_UP && L.cbsz != 16) return VS_ESHAPE;
                grp.p[j] = L.p;
                grp.p[j].ws = (float*)(ws + L.ws_off);
                grp.p[j].variant = g3v_of(L.cbsz, L.kind, L.p.mp != 0, L.big);
                lds = std::max(lds, g3v_lds(grp.p[j].variant));
                grp.wg_start[j] = (int)wg;
                wg += (long long)L.p.mbn * L.p.cbn * L.p.ksplit;
            }
            if (wg >= 2147483647ll) return VS_ESHAPE;
            for (int j = grp.n; j <= G3_GROUP_MAX; ++j) grp.wg_start[j] = (int)wg;
            rc = f16 ? g3b_uber_run<vs_half>(grp, lds, st) : g3b_uber_run<unsigned short>(grp, lds, st);
            if (rc) return rc;
        }
    } else
    for (int bucket = 0; bucket < 8; ++bucket) {
        const int cbsz = (bucket & 1) ? 8 : 16, kind = bucket < 2 || bucket >= 6 ? VS_CONV_K3 : (bucket < 4 ? VS_CONV_K2S2 : VS_CONV_UP);
        const bool mpb = bucket >= 6;                    // the M-packed 3x3x3 layers (8 stored P channels): kernel instantiations of their own
        std::vector<int> idx;
        for (int i = 0; i < count; ++i)
            if (plan.layers[i].cbsz == cbsz && plan.layers[i].kind == kind && (plan.layers[i].p.mp != 0) == mpb) idx.push_back(i);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return plan.layers[a].work > plan.layers[b].work; });
        for (size_t at = 0; at < idx.size(); at += G3_GROUP_MAX) {
            G3Group grp{};
            const int xcd_walk = vs_cfg().wgrad_xcd;
            grp.xcd = xcd_walk;
            grp.n = (int)std::min<size_t>(G3_GROUP_MAX, idx.size() - at);
            long long wg = 0;
            for (int j = 0; j < grp.n; ++j) {
                MultiLayer& L = plan.layers[idx[at + j]];
                grp.p[j] = L.p;
                grp.p[j].ws = (float*)(ws + L.ws_off);
                grp.wg_start[j] = (int)wg;
                wg += (long long)L.p.mbn * L.p.cbn * L.p.ksplit;
            }
            if (wg >= 2147483647ll) return VS_ESHAPE;
            for (int j = grp.n; j <= G3_GROUP_MAX; ++j) grp.wg_start[j] = (int)wg;
            if (kind == VS_CONV_UP) {
                if (cbsz != 16) return VS_ESHAPE;
                rc = f16 ? g3b_group_run<vs_half, 16, G3_UP>(grp, st) : g3b_group_run<unsigned short, 16, G3_UP>(grp, st);
            } else if (mpb) {
                if (f16) rc = cbsz == 16 ? g3b_group_run<vs_half, 16, G3_K3, true>(grp, st) : g3b_group_run<vs_half, 8, G3_K3, true>(grp, st);
                else rc = cbsz == 16 ? g3b_group_run<unsigned short, 16, G3_K3, true>(grp, st) : g3b_group_run<unsigned short, 8, G3_K3, true>(grp, st);
            } else if (f16) {
                if (kind == VS_CONV_K3) rc = cbsz == 16 ? g3b_group_run<vs_half, 16, G3_K3>(grp, st) : g3b_group_run<vs_half, 8, G3_K3>(grp, st);
                else rc = cbsz == 16 ? g3b_group_run<vs_half, 16, G3_K2S2>(grp, st) : g3b_group_run<vs_half, 8, G3_K2S2>(grp, st);
            } else {
                if (kind == VS_CONV_K3) rc = cbsz == 16 ? g3b_group_run<unsigned short, 16, G3_K3>(grp, st) : g3b_group_run<unsigned short, 8, G3_K3>(grp, st);
                else rc = cbsz == 16 ? g3b_group_run<unsigned short, 16, G3_K2S2>(grp, st) : g3b_group_run<unsigned short, 8, G3_K2S2>(grp, st);
            }
            if (rc) return rc;
        }
    }
    // ---- bias partials (in the all-buckets grid above when it is on) ----
    {
        std::vector<int> idx;
        for (int i = 0; i < count; ++i) if (descs[i].bias_g && !plan.layers[i].bias_fold && (!uber || descs[i].bias_rows >= 2147483647ll)) idx.push_back(i);
        for (size_t at = 0; at < idx.size(); at += G3_BIAS_MAX) {
            G3BiasGroup grp{};
            grp.n = (int)std::min<size_t>(G3_BIAS_MAX, idx.size() - at);
            int blk = 0;
            for (int j = 0; j < grp.n; ++j) {
                const int i = idx[at + j];
                const vs_wgrad_desc& d = descs[i];
                grp.d[j] = G3BiasDesc{d.bias_g, (double*)(ws + plan.layers[i].bias_off), d.bias_rows, d.bias_c_ch,
                                      d.bias_c_real, plan.layers[i].bias_nblk, 0};      // parts of one bias gradient: adjacent partial regions
                grp.blk_start[j] = blk;
                blk += plan.layers[i].bias_nblk;
            }
            for (int j = grp.n; j <= G3_BIAS_MAX; ++j) grp.blk_start[j] = blk;
            if (f16) hipLaunchKernelGGL(bias_partial_group_kernel<vs_half>, dim3(blk), dim3(256), 0, st, grp);
            else hipLaunchKernelGGL(bias_partial_group_kernel<unsigned short>, dim3(blk), dim3(256), 0, st, grp);
            VS_CHECK_LAUNCH();
        }
    }
    // ---- every reduction in one grid (G3_RED_MAX entries per launch) ----
    {
        std::vector<G3RedDesc> red;
        std::vector<int> blocks;
        for (int i = 0; i < count; ++i) {
            const MultiLayer& L = plan.layers[i];
            if (L.primary == i) {                        // one reduction per gradient, over the slabs of all its parts
                const long long slab_elems = (long long)L.p.mbn * L.p.cbn * L.ncb * 256;
                // a partition sums up to 8 slabs in one round of independent loads: no more partitions (= threads, waves) than that needs
                int parts = 1;
                while (parts < G3_RED_ROWS && parts * 8 < L.total_slabs) parts *= 2;
                red.push_back(G3RedDesc{(const float*)(ws + L.ws_off), L.dw, L.m_real, L.c_real, L.p.mbn, L.p.cbn, L.total_slabs, L.cbsz | (L.p.mp ? 0x100 : 0),
                                        L.kind != VS_CONV_K2S2 ? 27 : 8, L.ncb, 0, parts, L.swap});
                blocks.push_back(vs_ceil_div(slab_elems, 256 * (G3_RED_ROWS / parts)));
            }
            if (descs[i].bias_g && L.bias_primary == i) {
                red.push_back(G3RedDesc{(const float*)(ws + L.bias_off), descs[i].db, 0, descs[i].bias_c_real, 0, 0, L.bias_total_blk, 0, 0, 0, 1, G3_RED_ROWS});
                blocks.push_back(vs_ceil_div(descs[i].bias_c_real, 64));
            }
        }
        for (const vs_wgrad_desc& sd : slab_descs) {      // the layers whose slabs a backward-data launch wrote
            G3RedDesc r;
            int nb = 0;
            slab_red_entry(sd, r, nb);
            red.push_back(r);
            blocks.push_back(nb);
            if (sd.bias_g) { slab_bias_entry(sd, r, nb); red.push_back(r); blocks.push_back(nb); }
        }
        for (size_t at = 0; at < red.size(); at += G3_RED_MAX) {
            G3RedGroup grp{};
            grp.n = (int)std::min<size_t>(G3_RED_MAX, red.size() - at);
            long long blk = 0;
            for (int j = 0; j < grp.n; ++j) {
                grp.d[j] = red[at + j];
                grp.blk_start[j] = (int)blk;
                blk += blocks[at + j];
            }
            if (blk >= 2147483647ll) return VS_ESHAPE;
            for (int j = grp.n; j <= G3_RED_MAX; ++j) grp.blk_start[j] = (int)blk;
            hipLaunchKernelGGL(g3_reduce_group_kernel, dim3((unsigned)blk), dim3(64 * G3_RED_ROWS), 0, st, grp);
            VS_CHECK_LAUNCH();
        }
    }
    return VS_OK;
}
