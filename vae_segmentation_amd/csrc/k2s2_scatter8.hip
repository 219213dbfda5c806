// Backward-data of the stride-2 conv at full resolution (Down1: Conv3d(C, C, 2, stride 2), C = 8, joint_model.py:130; autograd of it), 16-bit storage:
//     gx[n][2v + t][m] = sum_c W[c][m][t] * gy[n][v][c]        (8 taps t, C -> C channels)
// with the fused InstanceNorm+ReLU-backward sums of gx against the conv's raw input (as g1_kernel<.., G1_PW, EPI_SCATTER> computes them).
//
// 64 FLOPs per output voxel against 32 bytes of traffic: pure streaming.  g1_kernel runs it as a 64-row MFMA tile per 256 coarse voxels — 864 one-tile
// workgroups at 96^3, each paying tables, one dependent load, an MFMA on 8 of 32 k-values and 16 scattered 8-byte stores per lane: 27 us for 60 MB
// (2.2 TB/s; 87 us at 160^3), half the rate of the k3t family.  Here a wave takes 32 (C = 16: 16) consecutive coarse voxels, lane = (coarse voxel, x parity [, channel half]): for each of the four
// (dz, dy) the 64 lanes read the mask and write gx as ONE contiguous KiB of a fine row; the 8 x 8 products per output voxel run on the vector ALU
// with the weights read from LDS (two distinct addresses per instruction: broadcast).  Persistent waves, sums reduced once per sample.
#include <stdlib.h>
#include <algorithm>
#include "igemm.h"

// LDS: weights float [8 taps][CH m][CH c], padded so that the lanes of one read (they differ in tap parity and channel half only) hit different banks;
// then mean / rstd of the mask [N * CH] each, then the waves' partial sums float [4][N][2 CH]
template <int CH> struct K2S8 {
    static constexpr int F = CH / 8;                      // 16-byte fragments per voxel
    static constexpr int MROW = CH + 4, TROW = CH * MROW + 8;
    static constexpr int W_FLOATS = 8 * TROW;
};

template <typename T, int CH>
__global__ __launch_bounds__(256) void k2s2_scatter8_kernel(const G1Params p, int items_per_sample, int total_items) {
    using L = K2S8<CH>;
    constexpr int F = L::F, VPI = 32 / F;                // coarse voxels per item (one wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w = (float*)smem;
    float* s_mean = s_w + L::W_FLOATS;
    float* s_rstd = s_mean + p.N * CH;
    float* s_part = s_rstd + p.N * CH;                   // [wave][n][2 CH]: every (wave, sample) slot is written at most once, summed in a fixed order at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * p.N * 2 * CH; i += 256) s_part[i] = 0.f;
    // weights: the packed scatter image [row block][lane][8]: row = t * CH + m at (row >> 4, lane = (row & 15) + 16 (c >> 3)), element c & 7
    for (int i = tid; i < 8 * CH * CH; i += 256) {
        const int r = i / CH, c = i - r * CH;
        const T* wp = (const T*)p.wp + (((r >> 4) * 64 + (r & 15) + 16 * (c >> 3)) * 8 + (c & 7));
        const unsigned int bits = (unsigned int)(*(const unsigned short*)wp);
        const int t = r / CH, m = r - t * CH;
        s_w[t * L::TROW + m * L::MROW + c] = H16<T>::lo(bits);
    }
    for (int i = tid; i < p.N * CH; i += 256) {
        float m, r;
        stats_to_mean_rstd_fast(p.mask_stats, (size_t)i, (size_t)p.N * CH, p.inv_count_out, p.eps, m, r);
        s_mean[i] = m; s_rstd[i] = r;
    }
    __syncthreads();

    const int S = p.D * p.H * p.W;                       // coarse voxels per sample
    const int HW = p.H * p.W;
    const int half = lane % F, dx = (lane / F) & 1, vl = lane / (2 * F);      // this lane: output channels 8 half .., x parity, coarse voxel of the item
    const int FW = 2 * p.W, FH = 2 * p.H;
    constexpr int VB = CH * 2;                           // bytes per voxel
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * S * VB));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * S * 8 * VB));
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * S * 8 * VB));

    float ssum[8], ssq[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { ssum[m] = 0.f; ssq[m] = 0.f; }
    int cur_n = -1;
    auto flush = [&](int n) {                             // this wave's sums of sample n -> its LDS slot (wave-uniform call; a wave meets a sample once)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            float s = ssum[m], q = ssq[m];
#pragma unroll
            for (int off = F; off < 64; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }      // lanes of equal channel half
            if (lane < F) {
                s_part[(wave * p.N + n) * 2 * CH + 8 * half + m] = s;
                s_part[(wave * p.N + n) * 2 * CH + CH + 8 * half + m] = q;
            }
            ssum[m] = 0.f; ssq[m] = 0.f;
        }
    };

    // workgroup b takes the contiguous run of items [b K, (b + 1) K): it meets few samples, its waves interleave inside the run
    const int K = (total_items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item_end = min(total_items, ((int)blockIdx.x + 1) * K);
    for (int item = (int)blockIdx.x * K + wave; item < item_end; item += 4) {
        const int n = item / items_per_sample, chunk = item - n * items_per_sample;
        if (n != cur_n) { if (cur_n >= 0) flush(cur_n); cur_n = n; }
        const int v = chunk * VPI + vl;                   // coarse voxel within the sample
        const bool ok = v < S;
        const int z = v / HW;                             // (z, y, x) of the coarse voxel: two divisions per 2 KiB of traffic
        const int r2 = v - z * HW;
        const int y = r2 / p.W;
        const int x = r2 - y * p.W;
        float gc[CH];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const u32x4 gq = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? (n * S + v) * VB + f * 16 : -1, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) { gc[8 * f + 2 * i] = H16<T>::lo(gq[i]); gc[8 * f + 2 * i + 1] = H16<T>::hi(gq[i]); }
        }
        // this lane's 16 bytes of the four fine voxels (2z + dz, 2y + dy, 2x + dx)
        int foff[4];
        u32x4 mk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dz = k >> 1, dy = k & 1;
            foff[k] = ok ? ((((n * 2 * p.D + 2 * z + dz) * FH + 2 * y + dy) * FW + 2 * x + dx) * VB + half * 16) : -1;
            mk[k] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(mrsrc, foff[k], 0, 0));
        }
        float mm[8], mr[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { mm[m] = s_mean[n * CH + 8 * half + m]; mr[m] = s_rstd[n * CH + 8 * half + m]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* wt = s_w + (2 * k + dx) * L::TROW + (8 * half) * L::MROW;     // tap (dz, dy, dx), this lane's eight output channels
            float o[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                float a = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < CH / 4; ++c4) {
                    const f32x4 wv = *(const f32x4*)(wt + m * L::MROW + 4 * c4);
                    a = fmaf(wv[0], gc[4 * c4], a); a = fmaf(wv[1], gc[4 * c4 + 1], a); a = fmaf(wv[2], gc[4 * c4 + 2], a); a = fmaf(wv[3], gc[4 * c4 + 3], a);
                }
                o[m] = a;
            }
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 pr;
                pr[0] = o[2 * i]; pr[1] = o[2 * i + 1];
                pk[i] = H16<T>::pack2(pr);
            }
            vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pk), yrsrc, foff[k], 0, 0);
            if (ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v0 = H16<T>::lo(pk[i]), v1 = H16<T>::hi(pk[i]);      // the stored (rounded) values, as g1_kernel sums them
                    const float x0 = H16<T>::lo(mk[k][i]), x1 = H16<T>::hi(mk[k][i]);
                    const float h0 = (x0 - mm[2 * i]) * mr[2 * i], h1 = (x1 - mm[2 * i + 1]) * mr[2 * i + 1];
                    const float g0 = h0 > 0.f ? v0 : 0.f, g1 = h1 > 0.f ? v1 : 0.f;
                    ssum[2 * i] += g0; ssq[2 * i] += g0 * h0;
                    ssum[2 * i + 1] += g1; ssq[2 * i + 1] += g1 * h1;
                }
            }
        }
    }
    if (cur_n >= 0) flush(cur_n);
    __syncthreads();
    // one atomic per (sample this workgroup met, statistic): 65 k atomics on 32 addresses — one per wave — made the first version 7x slower than the MFMA kernel
    for (int i = tid; i < p.N * 2 * CH; i += 256) {
        const int n = i / (2 * CH), k = i - n * 2 * CH;
        const float a0 = s_part[(0 * p.N + n) * 2 * CH + k], a1 = s_part[(1 * p.N + n) * 2 * CH + k], a2 = s_part[(2 * p.N + n) * 2 * CH + k],
                    a3 = s_part[(3 * p.N + n) * 2 * CH + k];
        if (a0 != 0.f || a1 != 0.f || a2 != 0.f || a3 != 0.f)
            stat_add(p.sums, (size_t)n * CH + (k % CH), (size_t)p.N * CH, k / CH, (double)a0 + (double)a1 + (double)a2 + (double)a3);
    }
}

// p as scatter_impl (conv_api.hip) fills it: x = gy on the coarse grid (N, D, H, W, C), y / mask_x on the fine grid, inv_count_out = 1 / fine voxels per sample
template <typename T, int CH>
static int k2s8_go(const G1Params& p, hipStream_t stream) {
    using L = K2S8<CH>;
    const long long S = (long long)p.D * p.H * p.W;
    const int vpi = 32 / L::F;
    const int items_per_sample = (int)((S + vpi - 1) / vpi);
    const long long total = (long long)items_per_sample * p.N;
    if (total >= 2147483647ll) return VS_ESHAPE;
    const int per_cu = vs_cfg().k2s8_wgs_per_cu;
    const long long cap = 256ll * per_cu;
    const int gx = (int)std::min<long long>((total + 3) / 4, cap);
    const size_t lds = ((size_t)L::W_FLOATS + (size_t)(2 * CH + 4 * 2 * CH) * p.N) * sizeof(float);
    if (lds > 64 * 1024) return VS_ESHAPE;
    hipLaunchKernelGGL((k2s2_scatter8_kernel<T, CH>), dim3(gx), dim3(256), lds, stream, p, items_per_sample, (int)total);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// (The kernel is written for C = 8 and 16; the 16-channel form — Down2's 48^3 gradient, 15 MB — measured 6 us per step SLOWER than g1_kernel's 26 us launch
// (2.3945 vs 2.4006 ms same box, profiles/r05_ab_k2s2_stream_joint96.json; standalone 42.1 vs 40.3 us at 2 x 48^3 coarse, 11.6 vs 10.0 at 24^3: 128 multiply-adds per
// 16 bytes written make it ALU-bound where the 8-channel form — 17.9 vs 21.7 us at 2 x 48^3, 30.2 vs 45.7 at 80^3 — streams) and is not instantiated.)
int k2s2_scatter8_launch(const G1Params& p, int dtype, hipStream_t stream) {
    if (p.C != 8 || p.M != p.C || !p.sums || !p.mask_x || !p.mask_stats || p.bias || p.x_stats) return VS_ESHAPE;
    if (dtype != VS_BF16 && dtype != VS_F16) return VS_EDTYPE;
    const long long S = (long long)p.D * p.H * p.W;
    if (S >= (1ll << 24) || (long long)p.N * S * 8 * p.C * 2 >= 2147483648ll) return VS_ESHAPE;
    return dtype == VS_BF16 ? k2s8_go<unsigned short, 8>(p, stream) : k2s8_go<vs_half, 8>(p, stream);
}
