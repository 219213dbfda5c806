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

enum { G3_K3 = 0, G3_K2S2 = 1 };
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
};

template <int CB, int KIND> struct G3Geo {
    static constexpr int NTAPS = KIND == G3_K3 ? 27 : 8;
    static constexpr int NCB = CB == 16 ? NTAPS : (NTAPS + 1) / 2;
    static constexpr int QZ = KIND == G3_K3 ? 6 : 8, QY = KIND == G3_K3 ? 6 : 8, QX = KIND == G3_K3 ? 18 : 32;
    static constexpr int QV = QZ * QY * QX;
};

#define G3_LDS_STATS 0                 // 4 x float[G3_MAXN][16]
#define G3_LDS_P (4 * G3_MAXN * 16 * 4)
#define G3_LDS_Q (G3_LDS_P + 256 * 16 * 4)


// Sum the four waves' accumulators in a fixed order (w0+w1+w2+w3: bitwise reproducible) through LDS and write ONE
// partial slab per workgroup: ws[ks][mb*cbn+cb][k][col][row].  `s_buf` needs NCB*256 floats.
template <int NCB>
__device__ __forceinline__ void g3_finish(f32x4 (&acc)[NCB], float* s_buf, float* ws_tile, int wave, int col, int g) {
    float* mine = s_buf + col * 16 + 4 * g;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
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

template <typename T, int CB, int KIND>
__global__ __launch_bounds__(256) void g3_kernel(const G3Params p) {
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
    const int mb = blockIdx.x / p.cbn, cb = blockIdx.x - mb * p.cbn;
    const int ks = blockIdx.y;
    const T* __restrict__ Pp = (const T*)p.P;
    const T* __restrict__ Qp = (const T*)p.Q;
    const bool p_stats = p.P_stats != nullptr, q_stats = p.Q_stats != nullptr;

    // mean / rstd tables for this WG's channel blocks, all samples
    for (int i = tid; i < p.N * 16; i += 256) {
        const int n = i >> 4, c = i & 15;
        float m = 0.f, r = 1.f;
        const int pc = mb * 16 + c;
        if (p_stats && pc < p.Mch) stats_to_mean_rstd(p.P_stats + ((size_t)n * p.Mch + pc) * 2, p.inv_cnt_p, p.eps, m, r);
        s_pm[i] = m; s_pr[i] = r;
        m = 0.f; r = 1.f;
        const int qc = cb * CB + c;
        if (q_stats && c < CB && qc < p.Cch) stats_to_mean_rstd(p.Q_stats + ((size_t)n * p.Cch + qc) * 2, p.inv_cnt_q, p.eps, m, r);
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

    for (int t = ks; t < p.total_tiles; t += p.ksplit) {
        const int n = t / p.tiles_per_sample;
        const int tl = t - n * p.tiles_per_sample;
        const int tx = tl % p.txn, ty = (tl / p.txn) % p.tyn, tz = tl / (p.txn * p.tyn);
        const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
        __syncthreads();      // previous tile fully consumed (also orders the stats tables on the first pass)
        // ---- stage P: 256 voxels x 16 channels of the m-block, fp32 in LDS ----
        for (int u = tid; u < 256 * (16 / EPL); u += 256) {
            const int v = u / (16 / EPL), part = u - v * (16 / EPL);
            const int lx = v & 15, ly = (v >> 4) & 3, lz = v >> 6;
            const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
            const int c0 = mb * 16 + part * EPL;
            float f[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) f[j] = 0.f;
            if (gz < p.Dp && gy < p.Hp && gx < p.Wp && c0 < p.Mch) {
                const size_t e = ((((size_t)n * p.Dp + gz) * p.Hp + gy) * p.Wp + gx) * p.Mch + c0;
                frag_unpack(*(const u32x4*)(Pp + e), f, (T*)nullptr);
                if (p_stats) {
#pragma unroll
                    for (int j = 0; j < EPL; ++j) {
                        const float tt = (f[j] - s_pm[n * 16 + part * EPL + j]) * s_pr[n * 16 + part * EPL + j];
                        f[j] = tt > 0.f ? tt : 0.f;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; j += 4)
                *(f32x4*)(s_p + v * 16 + part * EPL + j) = f32x4{f[j], f[j + 1], f[j + 2], f[j + 3]};
        }
        // ---- stage Q: halo (K3) or 2x-upsampled (K2S2) region x CB channels ----
        for (int u = tid; u < QV * (CB / EPL); u += 256) {
            const int v = u / (CB / EPL), part = u - v * (CB / EPL);
            const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
            int gz, gy, gx;
            if (KIND == G3_K3) { gz = z0 + lz - 1; gy = y0 + ly - 1; gx = x0 + lx - 1; }
            else { gz = 2 * z0 + lz; gy = 2 * y0 + ly; gx = 2 * x0 + lx; }
            const int c0 = cb * CB + part * EPL;
            float f[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) f[j] = 0.f;
            if (gz >= 0 && gz < p.Dq && gy >= 0 && gy < p.Hq && gx >= 0 && gx < p.Wq && c0 < p.Cch) {
                const size_t e = ((((size_t)n * p.Dq + gz) * p.Hq + gy) * p.Wq + gx) * p.Cch + c0;
                frag_unpack(*(const u32x4*)(Qp + e), f, (T*)nullptr);
                if (q_stats) {
#pragma unroll
                    for (int j = 0; j < EPL; ++j) {
                        const float tt = (f[j] - s_qm[n * 16 + part * EPL + j]) * s_qr[n * 16 + part * EPL + j];
                        f[j] = tt > 0.f ? tt : 0.f;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; j += 4)
                *(f32x4*)(s_q + v * CB + part * EPL + j) = f32x4{f[j], f[j + 1], f[j + 2], f[j + 3]};
        }
        __syncthreads();
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
    g3_finish<NCB>(acc, s_p, p.ws + (size_t)ks * slab_elems + ((size_t)blockIdx.x * NCB) * 256, wave, col, g);
}

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

__device__ __forceinline__ bf16x8 tr_pair(const char* s_base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_base + off1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

#define G3B_LDS_P (4 * G3_MAXN * 16 * 4)
#define G3B_LDS_Q (G3B_LDS_P + 256 * 16 * 2)

// Tile t+ksplit's P and Q fragments are requested (bounds-checked buffer loads: out-of-volume fragments read as zero,
// no branches) right after tile t went to LDS, so their latency is covered by tile t's MFMA phase; the lazy operands'
// normalise+ReLU runs as packed fma / packed max (common.h act8).
template <int CB, int KIND>
__global__ __launch_bounds__(256) void g3b_kernel(const G3Params p) {
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
    const int mb = blockIdx.x / p.cbn, cb = blockIdx.x - mb * p.cbn;
    const int ks = blockIdx.y;
    const bool p_stats = p.P_stats != nullptr, q_stats = p.Q_stats != nullptr;
    const i32x4 prsrc = make_rsrc(p.P, (unsigned int)((long long)p.N * p.Dp * p.Hp * p.Wp * p.Mch * 2));
    const i32x4 qrsrc = make_rsrc(p.Q, (unsigned int)((long long)p.N * p.Dq * p.Hq * p.Wq * p.Cch * 2));

    // ---- tile-independent fragment geometry ----
    // P fragment b: voxel v = (tid + 256 b) >> 1 of the 4x4x16 tile, channel half `ppart`; Q fragment b: voxel qv of the halo /
    // strided region, 16-byte part `qpart` (constant per thread: 256 % QU == 0)
    const int ppart = tid & 1, qpart = tid % QU;
    const bool pch_ok = mb * 16 + ppart * 8 < p.Mch, qch_ok = cb * CB + qpart * 8 < p.Cch;
    int prel[2], pzyx[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int v = (tid + b * 256) >> 1;
        const int lx = v & 15, ly = (v >> 4) & 3, lz = v >> 6;
        prel[b] = (((lz * p.Hp + ly) * p.Wp + lx) * p.Mch + mb * 16 + ppart * 8) * 2;
        pzyx[b] = pch_ok ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    int qrel[NITQ], qzyx[NITQ];
#pragma unroll
    for (int b = 0; b < NITQ; ++b) {
        const int u = tid + b * 256;
        const int v = u / QU;
        const int lx = v % QX, ly = (v / QX) % QY, lz = v / (QX * QY);
        qrel[b] = (((lz * p.Hq + ly) * p.Wq + lx) * p.Cch + cb * CB + qpart * 8) * 2;
        qzyx[b] = (u < NQ && qch_ok) ? (lz | (ly << 8) | (lx << 16)) : 0x00ffffff;
    }
    u32x4 pv[2], qv[NITQ];
    unsigned int okbits = 0;                     // bit b: P fragment b inside the volume; bit 2 + b: Q fragment b
    auto request = [&](int t) {
        const int n = t / p.tiles_per_sample;
        const int tl = t - n * p.tiles_per_sample;
        const int tx = tl % p.txn, ty = (tl / p.txn) % p.tyn, tz = tl / (p.txn * p.tyn);
        const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
        okbits = 0;
        const int pbase = (((n * p.Dp + z0) * p.Hp + y0) * p.Wp + x0) * p.Mch * 2;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int gz = z0 + (pzyx[b] & 0xff), gy = y0 + ((pzyx[b] >> 8) & 0xff), gx = x0 + (pzyx[b] >> 16);
            const bool ok = gz < p.Dp && gy < p.Hp && gx < p.Wp;
            okbits |= ok ? (1u << b) : 0u;
            pv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(prsrc, ok ? pbase + prel[b] : -1, 0, 0));
        }
        const int qz0 = KIND == G3_K3 ? z0 - 1 : 2 * z0, qy0 = KIND == G3_K3 ? y0 - 1 : 2 * y0, qx0 = KIND == G3_K3 ? x0 - 1 : 2 * x0;
        const int qbase = (((n * p.Dq + qz0) * p.Hq + qy0) * p.Wq + qx0) * p.Cch * 2;
#pragma unroll
        for (int b = 0; b < NITQ; ++b) {
            const int gz = qz0 + (qzyx[b] & 0xff), gy = qy0 + ((qzyx[b] >> 8) & 0xff), gx = qx0 + (qzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.Dq && (unsigned)gy < (unsigned)p.Hq && (unsigned)gx < (unsigned)p.Wq;
            okbits |= ok ? (4u << b) : 0u;
            qv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(qrsrc, ok ? qbase + qrel[b] : -1, 0, 0));
        }
    };
    auto commit = [&](int n) {                   // registers -> (normalised) LDS tiles
        f32x2 sc[4], sh[4];
        if (p_stats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_psc + n * 16 + ppart * 8 + 2 * i);
                sh[i] = *(const f32x2*)(s_psh + n * 16 + ppart * 8 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            u32x4 v = pv[b];
            if (p_stats) {
                const u32x4 a = act8(v, sc, sh);
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
            if (q_stats) {
                const u32x4 a = act8(v, sc, sh);
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
        if (p_stats && pc < p.Mch) stats_to_mean_rstd_fast(p.P_stats + ((size_t)n * p.Mch + pc) * 2, p.inv_cnt_p, p.eps, m, r);
        s_psc[i] = r; s_psh[i] = -m * r;
        m = 0.f; r = 1.f;
        const int qc = cb * CB + c;
        if (q_stats && c < CB && qc < p.Cch) stats_to_mean_rstd_fast(p.Q_stats + ((size_t)n * p.Cch + qc) * 2, p.inv_cnt_q, p.eps, m, r);
        s_qsc[i] = r; s_qsh[i] = -m * r;
    }

    // per-lane byte offsets into the Q tile of this lane's tr-read row for each column block (tap part only)
    int qoff[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        int tap = CB == 16 ? k : 2 * k + (p4 >> 1);
        if (tap >= NTAPS) tap = KIND == G3_K3 ? 13 : 0;
        int dz, dy, dx;
        if (KIND == G3_K3) { dz = tap / 9; dy = (tap / 3) % 3; dx = tap % 3; }
        else { dz = (tap >> 2) & 1; dy = (tap >> 1) & 1; dx = tap & 1; }
        qoff[k] = ((dz * QY + dy) * QX + dx) * QROW + (CB == 16 ? p4 * 8 : (p4 & 1) * 8);
    }

    f32x4 acc[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (; t < p.total_tiles; t += p.ksplit) {
        __syncthreads();                         // tables visible / every wave is done reading the previous tile
        commit(t / p.tiles_per_sample);
        __syncthreads();
        if (t + p.ksplit < p.total_tiles) request(t + p.ksplit);
        // ---- two K-steps of 32 voxels: y rows (2s, 2s+1) of this wave's z-slice ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int xr = 4 * g + q4;                                   // this lane's tr-read row: voxel x
            const int pa0 = (((wave * 4 + 2 * s) * 16 + xr) * 32) + p4 * 8;
            const bf16x8 a = tr_pair(s_p, pa0, pa0 + 16 * 32);
            int qb0, qb1;
            if (KIND == G3_K3) {
                qb0 = ((wave * QY + 2 * s) * QX + xr) * QROW;
                qb1 = qb0 + QX * QROW;
            } else {
                qb0 = ((2 * wave * QY + 2 * (2 * s)) * QX + 2 * xr) * QROW;
                qb1 = qb0 + 2 * QX * QROW;
            }
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                const bf16x8 b = tr_pair(s_q, qb0 + qoff[k], qb1 + qoff[k]);
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
            }
        }
    }

    const size_t slab_elems = (size_t)p.mbn * p.cbn * NCB * 256;
    g3_finish<NCB>(acc, (float*)s_p, p.ws + (size_t)ks * slab_elems + ((size_t)blockIdx.x * NCB) * 256, wave, col, g);
}

// Sum the partial slabs (fixed order, fp64) into the reference's [m][c][tap] layout.  A block = 64 consecutive slab
// elements (coalesced reads) x 16 slab partitions; each thread keeps 8 loads in flight, LDS combines the partitions.
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
    const int ntaps = kind == VS_CONV_K3 ? 27 : 8;
    ncb = cbsz == 16 ? ntaps : (ntaps + 1) / 2;
    tyn = (hp + 3) / 4; txn = (wp + 15) / 16;
    tiles_per_sample = ((dp + 3) / 4) * tyn * txn;
    const long long total = (long long)tiles_per_sample * n;
    static const long long wg_target = getenv("VS_WGRAD_WGS") ? atoll(getenv("VS_WGRAD_WGS")) : 512;   // tuning knob: ~2 workgroups per CU overall
    long long want = (wg_target + (long long)mbn * cbn - 1) / ((long long)mbn * cbn);
    // fp32 (parity) mode: at most two tiles per workgroup, so an fp32 MFMA accumulator never chains more than 128
    // products before the fp64 slab reduction
    if (short_chains && (total + 1) / 2 > want) want = (total + 1) / 2;
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

template <int CB, int KIND>
static int g3b_run(const G3Params& p, float* dw, int m_real, int c_real, hipStream_t s) {
    using GEO = G3Geo<CB, KIND>;
    constexpr size_t lds = G3B_LDS_Q + (size_t)GEO::QV * CB * 2;
    auto kern = g3b_kernel<CB, KIND>;
    if (lds > 64 * 1024) {
        static const hipError_t attr_err =
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr_err != hipSuccess) return (int)attr_err;
    }
    hipLaunchKernelGGL(kern, dim3(p.mbn * p.cbn, p.ksplit), dim3(256), lds, s, p);
    VS_CHECK_LAUNCH();
    const long long slab_elems = (long long)p.mbn * p.cbn * GEO::NCB * 256;
    hipLaunchKernelGGL((g3_reduce_kernel<CB, KIND>), dim3(vs_ceil_div(slab_elems, 64)), dim3(1024), 0, s, p.ws, dw, m_real,
                       c_real, p.mbn, p.cbn, p.ksplit);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T, int CB, int KIND>
static int g3_run(const G3Params& p, float* dw, int m_real, int c_real, hipStream_t s) {
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
    const long long slab_elems = (long long)p.mbn * p.cbn * GEO::NCB * 256;
    hipLaunchKernelGGL((g3_reduce_kernel<CB, KIND>), dim3(vs_ceil_div(slab_elems, 64)), dim3(1024), 0, s, p.ws, dw, m_real,
                       c_real, p.mbn, p.cbn, p.ksplit);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_conv_wgrad(const void* P, const double* p_stats, const void* Q, const double* q_stats, float* dw,
                             void* workspace, size_t workspace_bytes, int n, int dp, int hp, int wp, int m_ch, int c_ch,
                             int m_real, int c_real, int kind, int dtype, float eps, void* stream) {
    if (!P || !Q || !dw || !workspace) return VS_EINVAL;
    if (n <= 0 || n > G3_MAXN || dp <= 0 || hp <= 0 || wp <= 0) return VS_ESHAPE;
    if (m_ch % 8 || c_ch % 8 || m_real > m_ch || c_real > c_ch || m_real <= 0 || c_real <= 0) return VS_ESHAPE;
    if (kind != VS_CONV_K3 && kind != VS_CONV_K2S2) return VS_EINVAL;
    if (dtype != VS_F32 && dtype != VS_BF16) return VS_EDTYPE;
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
    p.eps = eps;
    p.inv_cnt_p = 1.0 / ((double)dp * hp * wp);
    p.inv_cnt_q = 1.0 / ((double)p.Dq * p.Hq * p.Wq);
    hipStream_t st = (hipStream_t)stream;
#define G3_GO(T) \
    if (kind == VS_CONV_K3) return cbsz == 16 ? g3_run<T, 16, G3_K3>(p, dw, m_real, c_real, st) : g3_run<T, 8, G3_K3>(p, dw, m_real, c_real, st); \
    return cbsz == 16 ? g3_run<T, 16, G3_K2S2>(p, dw, m_real, c_real, st) : g3_run<T, 8, G3_K2S2>(p, dw, m_real, c_real, st);
    if (dtype == VS_F32) { G3_GO(float) }
#undef G3_GO
    // g3b_kernel addresses P and Q with signed 32-bit byte offsets
    if ((long long)n * dp * hp * wp * m_ch * 2 >= 2147483648ll || (long long)n * p.Dq * p.Hq * p.Wq * c_ch * 2 >= 2147483648ll) return VS_ESHAPE;
    if (kind == VS_CONV_K3) return cbsz == 16 ? g3b_run<16, G3_K3>(p, dw, m_real, c_real, st) : g3b_run<8, G3_K3>(p, dw, m_real, c_real, st);
    return cbsz == 16 ? g3b_run<16, G3_K2S2>(p, dw, m_real, c_real, st) : g3b_run<8, G3_K2S2>(p, dw, m_real, c_real, st);
}

