// 3x3x3 (pad 1) implicit-GEMM convolution, forward and backward-data: persistent, register-prefetched.
// Instantiated for fp32 storage (the parity mode); bf16 storage runs k3b_kernel (igemm_k3b.h).
//
// Same GEMM orientation, fragment formats, LDS halo tile and epilogues as g1_kernel (igemm.h), restructured for the
// fact that on this op a workgroup is latency-bound, not MFMA-bound (a 4x4x16 tile is ~450 MFMA cycles against
// microseconds of global-load latency): a fixed grid of workgroups walks the tile list (tile index strided by the grid,
// so concurrently processed tiles are spatial neighbours and share halos in L2), and while tile i is multiplied out of
// LDS the global loads of stage i+1 (next channel chunk or next tile) are already in flight into registers
// (issue-early / write-late staging).  Per-(n,c) statistics are kept in registers across the tiles of one sample and
// flushed with one fp64 atomic per (row, statistic) when the sample changes.
#pragma once
#include <stdlib.h>
#include "igemm.h"

#define K3_LDS_RED 0          // float[4][64][2]
#define K3_LDS_TAPS 2048      // int[64]
#define K3_LDS_TILE 2304      // halo tile, then mean/rstd tables float[2][N*C]

// YT: rows of 16 voxels per wave (tile 4 x YT x 16).  The exact-f32 MFMA runs at 1/16 of the 16-bit rate, so the fp32 layers are bound by MFMA cycles
// per wave; at the 12^3 / 24^3 levels a 4x4x16 tiling yields 72-288 workgroups for 1024 SIMDs and one wave carries 4 column groups x all of K:
// shorter tiles (more halo, which is L2-resident there) spread the same MFMAs over 4x the waves.
// TY (fp32, 8 stored input and <= 8 output channels: the full-resolution layers): Toeplitz rows along y.  With 8 real rows half of every exact-f32
// MFMA multiplies padding, and these launches are MFMA-bound (96^3: 135 us against 80 us of MFMA cycles).  Rows become (dy2, co) — two output
// voxels adjacent in y times 8 channels — over a 4-tall window: 36 window taps (dz, wy, dx) instead of 27, weights W[co][ci][dz][wy - dy2][dx]
// or zero (pack.hip, vs_k3_toeplitz_f32), a wave's four rows of 16 voxels become two row PAIRS: 18 k-groups x 2 column groups x 4 MFMAs per tile
// and wave instead of 14 x 4 x 4 (1.56x fewer).
template <typename T, int CK, int MT, int EPI, int YT = 4, bool TY = false>
__global__ __launch_bounds__(256) void k3_kernel(const G1Params p) {
    using E = ET<T>;
    constexpr int EPL = E::EPL, KG = E::KG;
    static_assert(!TY || (sizeof(T) == 4 && CK == 8 && MT == 16 && YT == 4), "y-Toeplitz: the fp32 8-channel layers");
    constexpr int NTAPS = TY ? 36 : 27;
    constexpr int NCGW = TY ? 2 : YT;                    // column groups per wave (TY: row pairs)
    constexpr int NKG = (NTAPS * CK + KG - 1) / KG;
    constexpr int RB = MT / 16;
    constexpr int CKB = CK * (int)sizeof(T);
    constexpr int TPK = KG > CK ? KG / CK : 1;
    constexpr int KPT = KG > CK ? 1 : CK / KG;
    constexpr int U = CKB / 16;
    constexpr int PLANE = (YT + 2) * 18, TVOX = 6 * PLANE;     // staged halo voxels
    constexpr int NU = TVOX * U;
    constexpr int NIT = (NU + 255) / 256;
    constexpr bool PF = NIT <= 14;                       // prefetch the next stage into registers while computing
    constexpr int SB = PF ? NIT : (NIT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3_LDS_RED);
    int* s_taps = (int*)(smem + K3_LDS_TAPS);
    char* s_tile = smem + K3_LDS_TILE;
    float* s_mean = (float*)(s_tile + TVOX * CKB);
    float* s_rstd = s_mean + p.N * p.C;
    float* s_mkm = s_rstd + p.N * p.C;                   // mean / rstd of the mask tensor's channels (fused IN-bwd sums)
    float* s_mkr = s_mkm + p.N * p.M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    const bool has_stats = p.x_stats != nullptr;
    const T* __restrict__ xin = (const T*)p.x;
    const int total_tiles = p.tiles_per_sample * p.N;

    if (has_stats) {
        for (int i = tid; i < p.N * p.C; i += 256) {
            float m, r;
            stats_to_mean_rstd(p.x_stats, (size_t)i, (size_t)p.N * p.C, p.inv_count_in, p.eps, m, r);
            s_mean[i] = m;
            s_rstd[i] = r;
        }
    }
    if (p.sums != nullptr) {
        for (int i = tid; i < p.N * p.M; i += 256) {
            float m, r;
            stats_to_mean_rstd(p.mask_stats, (size_t)i, (size_t)p.N * p.M, p.inv_count_out, p.eps, m, r);
            s_mkm[i] = m;
            s_mkr[i] = r;
        }
    }
    if (tid < 64) {
        if constexpr (TY) {
            const int t = tid < 36 ? tid : 16;            // (dz, wy, dx); padded entries read the centre voxel (their weights are zero)
            const int dz = t / 12, wy = (t / 3) % 4, dx = t % 3;
            s_taps[tid] = (dz * PLANE + wy * 18 + dx) * CKB;
        } else {
            const int t = tid < 27 ? tid : 13;
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            s_taps[tid] = (dz * PLANE + dy * 18 + dx) * CKB;
        }
    }
    int lds_base[NCGW];
#pragma unroll
    for (int cg = 0; cg < NCGW; ++cg) lds_base[cg] = (wave * PLANE + (TY ? 2 * cg : cg) * 18 + col) * CKB + ((g * EPL) % CK) * (int)sizeof(T);

    const u32x4* __restrict__ wp = (const u32x4*)p.wp;
    const size_t rb_stride = (size_t)p.nch * NKG * 64;

    // ---- staging helpers --------------------------------------------------------------------------------
    // The integer work of mapping a thread's fragments to tile voxels (divisions by 18 / 6 / U, 64-bit offsets) used to
    // cost ~80 VALU instructions per fragment per stage and made the kernel issue-bound (PMC: 10k VALU vs 0.9k MFMA per
    // wave on the 3^3 layers).  Everything tile-independent is computed once per thread: the fragment's offset relative
    // to the tile origin, its packed tile coordinates (for the bounds test) and its LDS byte offset.
    u32x4 vals[SB];
    bool ok[SB];
    int rel_off[SB], tzyx[SB], lds_w[SB];
    if constexpr (PF) {
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            const int u = tid + b * 256;
            const int tv = u / U, part = u - tv * U;
            const int tx_ = tv % 18, ty_ = (tv / 18) % (YT + 2), tz_ = tv / PLANE;
            rel_off[b] = ((tz_ * p.H + ty_) * p.W + tx_) * p.C + part * EPL;
            tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;      // out-of-list fragments fail every bounds test
            lds_w[b] = tv * CKB + part * 16;
        }
    }
    auto tile_origin = [&](int t, int& n, int& z0, int& y0, int& x0) {
        n = t / p.tiles_per_sample;
        const int tl = t - n * p.tiles_per_sample;
        x0 = (tl % p.txn) * 16;
        y0 = ((tl / p.txn) % p.tyn) * YT;
        z0 = (tl / (p.txn * p.tyn)) * 4;
    };
    auto stage_load = [&](int t, int ch, int it0) {
        int n, z0, y0, x0;
        tile_origin(t, n, z0, y0, x0);
        if constexpr (PF) {
            // element offset of tile voxel (0,0,0) = volume voxel (z0-1, y0-1, x0-1); fits 32 bits (checked on the host)
            const int base = (((n * p.D + z0 - 1) * p.H + y0 - 1) * p.W + x0 - 1) * p.C + ch * CK;
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int gz = z0 - 1 + (tzyx[b] & 0xff), gy = y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = x0 - 1 + (tzyx[b] >> 16);
                ok[b] = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const int e = ok[b] ? base + rel_off[b] : 0;
                vals[b] = *(const u32x4*)(xin + e);
            }
        } else {
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int u = tid + (it0 + b) * 256;
                const int tv = u / U, part = u - tv * U;
                const int tx_ = tv % 18, ty_ = (tv / 18) % (YT + 2), tz_ = tv / PLANE;
                const int gz = z0 + tz_ - 1, gy = y0 + ty_ - 1, gx = x0 + tx_ - 1;
                ok[b] = (it0 + b < NIT) && u < NU && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const size_t e = ok[b] ? ((((size_t)n * p.D + gz) * p.H + gy) * p.W + gx) * p.C + ch * CK + part * EPL : 0;
                vals[b] = *(const u32x4*)(xin + e);
            }
        }
    };
    auto stage_write = [&](int n, int ch, int it0) {
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            const int u = tid + (it0 + b) * 256;
            if (it0 + b < NIT && u < NU) {
                int lw, c0;
                if constexpr (PF) { lw = lds_w[b]; c0 = ch * CK + ((lw >> 4) % U) * EPL; }
                else { const int tv = u / U, part = u - tv * U; lw = tv * CKB + part * 16; c0 = ch * CK + part * EPL; }
                u32x4 val = u32x4{0u, 0u, 0u, 0u};
                if (ok[b]) {        // halo fragments outside the volume stay zero and skip the normalisation arithmetic
                    val = vals[b];
                    if (has_stats) val = act_transform<T, CK>(val, s_mean + n * p.C, s_rstd + n * p.C, c0);
                }
                *(u32x4*)(s_tile + lw) = val;
            }
        }
    };

    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }

    // (tried: keeping the single-chunk layers' weight fragments in registers — costs occupancy, 15 % slower)
    int t = blockIdx.x;
    __syncthreads();                                     // tables visible
    if (PF && t < total_tiles) stage_load(t, 0, 0);

    for (; t < total_tiles; t += gridDim.x) {
        int n, z0, y0, x0;
        tile_origin(t, n, z0, y0, x0);
        f32x4 acc[RB][NCGW];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < NCGW; ++cg) acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < p.nch; ++ch) {
            if constexpr (PF) {
                __syncthreads();                         // every wave is done reading the previous stage's tile
                stage_write(n, ch, 0);
                __syncthreads();
                if constexpr (KG > CK) {              // (few weight loads per tile: prefetch right away)
                    const bool more_ch = ch + 1 < p.nch;
                    const int tn = more_ch ? t : t + (int)gridDim.x;
                    if (tn < total_tiles) stage_load(tn, more_ch ? ch + 1 : 0, 0);
                }
            } else {
                __syncthreads();
#pragma unroll 1
                for (int it0 = 0; it0 < NIT; it0 += SB) {
                    stage_load(t, ch, it0);
                    stage_write(n, ch, it0);
                }
                __syncthreads();
            }
            const u32x4* wch = wp + (size_t)ch * NKG * 64 + lane;

            if constexpr (KG > CK) {
                const int sub = (g * EPL) / CK;
#pragma unroll
                for (int kg = 0; kg < NKG; ++kg) {
                    u32x4 a[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a[rb] = wch[(size_t)(rb0 + rb) * rb_stride + kg * 64];
                    const int toff = s_taps[kg * TPK + sub];
                    u32x4 b[NCGW];
#pragma unroll
                    for (int cg = 0; cg < NCGW; ++cg) b[cg] = *(const u32x4*)(s_tile + lds_base[cg] + toff);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cg = 0; cg < NCGW; ++cg) acc[rb][cg] = mfma16(a[rb], b[cg], acc[rb][cg], (T*)nullptr);
                }
            } else {
                constexpr int NK = NTAPS * KPT;
                constexpr int PD = RB <= 2 ? 9 : 3;          // prefetch distance in k-groups (27 was tried for RB=1: slower)
                u32x4 abuf[PD][RB];
#pragma unroll
                for (int j = 0; j < PD; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        abuf[j][rb] = j < NK ? wch[(size_t)(rb0 + rb) * rb_stride + j * 64] : u32x4{0u, 0u, 0u, 0u};
                // B fragments (LDS) are fetched one k-group ahead as well: with few waves per CU on the deep layers an
                // un-prefetched ds_read -> MFMA chain exposes the full LDS latency on every MFMA.
                auto lds_off = [&](int kg) {
                    const int tap = kg / KPT, kk = kg - tap * KPT;
                    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                    return (dz * PLANE + dy * 18 + dx) * CKB + kk * KG * (int)sizeof(T);
                };
                u32x4 bnext[YT];
#pragma unroll
                for (int cg = 0; cg < YT; ++cg) bnext[cg] = *(const u32x4*)(s_tile + lds_base[cg] + lds_off(0));
#pragma unroll 1
                for (int kgb = 0; kgb < NK; kgb += PD) {
#pragma unroll
                    for (int j = 0; j < PD; ++j) {
                        const int kg = kgb + j;
                        if (kg < NK) {
                            u32x4 a[RB], b[YT];
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb) a[rb] = abuf[j][rb];
#pragma unroll
                            for (int cg = 0; cg < YT; ++cg) b[cg] = bnext[cg];
                            if (kg + 1 < NK) {
                                const int o = lds_off(kg + 1);
#pragma unroll
                                for (int cg = 0; cg < YT; ++cg) bnext[cg] = *(const u32x4*)(s_tile + lds_base[cg] + o);
                            }
                            if (kg + PD < NK) {
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb) abuf[j][rb] = wch[(size_t)(rb0 + rb) * rb_stride + (kg + PD) * 64];
                            }
                            if constexpr (PF) {
                                // vmcnt retires loads in issue order: the next stage's activation loads are requested only
                                // after this chunk's last weight fragment, so no weight fragment ever waits behind them
                                if (kg + PD == NK || (NK <= PD && kg == 0)) {
                                    const bool more_ch = ch + 1 < p.nch;
                                    const int tn = more_ch ? t : t + (int)gridDim.x;
                                    if (tn < total_tiles) stage_load(tn, more_ch ? ch + 1 : 0, 0);
                                }
                            }
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                for (int cg = 0; cg < YT; ++cg) acc[rb][cg] = mfma16(a[rb], b[cg], acc[rb][cg], (T*)nullptr);
                        }
                    }
                }
            }
        }

        // ---- epilogue of this tile ----
        const int oz = z0 + wave;
        if constexpr (EPI == EPI_SOFTMAX2) {
            if (TY ? (g & 1) == 0 : g == 0) {                 // the lanes holding rows (dy2,) 0 .. 3: the two logits are rows 0 and 1
                const float b0 = p.bias ? p.bias[0] : 0.f, b1 = p.bias ? p.bias[1] : 0.f;
                const size_t V = (size_t)p.D * p.H * p.W;
#pragma unroll
                for (int cg = 0; cg < NCGW; ++cg) {
                    const int oy = TY ? y0 + 2 * cg + (g >> 1) : y0 + cg, ox = x0 + col;
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
                const int row = TY ? 4 * (g & 1) : (rb0 + rb) * 16 + 4 * g;      // TY: lane group g holds (dy2 = g >> 1, channels 4 (g & 1) ..)
                const bool rvalid = row < p.M;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && rvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = p.bias[row + r];
                }
#pragma unroll
                for (int cg = 0; cg < NCGW; ++cg) {
                    const int oy = TY ? y0 + 2 * cg + (g >> 1) : y0 + cg, ox = x0 + col;
                    if (!(rvalid && oz < p.D && oy < p.H && ox < p.W)) continue;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = E::rnd(acc[rb][cg][r] + bv[r]);
                    const size_t e = ((((size_t)n * p.D + oz) * p.H + oy) * p.W + ox) * p.M + row;
                    store4<T>(yout + e, v);
                    if (p.sums != nullptr) {
                        float xv[4];
                        if constexpr (sizeof(T) == 4) {
                            const f32x4 xx = *(const f32x4*)((const float*)p.mask_x + e);
                            xv[0] = xx[0]; xv[1] = xx[1]; xv[2] = xx[2]; xv[3] = xx[3];
                        } else {
                            const u32x2 xx = *(const u32x2*)((const T*)p.mask_x + e);
                            const unsigned int w2[2] = {xx[0], xx[1]};
                            widen4<T>(w2, xv);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (xv[r] - s_mkm[n * p.M + row + r]) * s_mkr[n * p.M + row + r];
                            const float gm = xh > 0.f ? v[r] : 0.f;
                            ssum[rb][r] += gm; ssq[rb][r] += gm * xh;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[rb][r] += v[r]; ssq[rb][r] += v[r] * v[r]; }
                    }
                }
            }
            double* const red_dst0 = p.sums != nullptr ? p.sums : p.y_stats;
            double* const red_dst = red_dst0;
            if (red_dst != nullptr) {
                const int tn = t + (int)gridDim.x;
                const bool flush = tn >= total_tiles || tn / p.tiles_per_sample != n;     // workgroup-uniform
                if (flush) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s = ssum[rb][r], q = ssq[rb][r];
                            { s = row16_sum(s); q = row16_sum(q); }      // DPP: the four-step __shfl_xor butterfly was four ds_bpermute round trips per statistic
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
                        if (TY ? lr < 8 && lr < p.M : row < p.M) {
                            double tot = (double)s_red[(0 * 64 + lr) * 2 + st] + (double)s_red[(1 * 64 + lr) * 2 + st] +
                                         (double)s_red[(2 * 64 + lr) * 2 + st] + (double)s_red[(3 * 64 + lr) * 2 + st];
                            if constexpr (TY) {             // channel lr: rows lr (dy2 = 0) and lr + 8 (dy2 = 1); lane row 4 g + r = 8 dy2 + channel
                                tot += (double)s_red[(0 * 64 + lr + 8) * 2 + st] + (double)s_red[(1 * 64 + lr + 8) * 2 + st] +
                                       (double)s_red[(2 * 64 + lr + 8) * 2 + st] + (double)s_red[(3 * 64 + lr + 8) * 2 + st];
                            }
                            stat_add(red_dst, (size_t)n * p.M + row, (size_t)p.N * p.M, st, tot);
                        }
                    }
                    __syncthreads();                     // s_red is reused by a later flush
                }
            }
        }
    }
}

template <typename T, int CK, int MT, int EPI, int YT = 4, bool TY = false>
static int k3_launch(const G1Params& p_in, int tiles_total, int row_tiles, hipStream_t stream) {
    G1Params p = p_in;
    if (YT != 4) {                                       // re-tile the volume in 4 x YT x 16 tiles
        p.tyn = (p.H + YT - 1) / YT;
        p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
        tiles_total = p.tiles_per_sample * p.N;
    }
    const size_t tables = (size_t)2 * p.N * p.C * sizeof(float) + (p.sums ? (size_t)2 * p.N * p.M * sizeof(float) : 0);
    const size_t lds = K3_LDS_TILE + (size_t)6 * (YT + 2) * 18 * CK * sizeof(T) + tables;
    if (lds > 160 * 1024) return VS_ESHAPE;
    auto kern = k3_kernel<T, CK, MT, EPI, YT, TY>;
    // idempotent one-time opt-in to the full 160 KiB of dynamic LDS (not a stream operation)
    static const hipError_t attr_err =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    // persistent grid: a few workgroups per CU, each walking a strided slice of the tile list
    const int per_cu = vs_cfg().k3_wgs_per_cu > 0 ? vs_cfg().k3_wgs_per_cu : 4;   // tuning knob (vs_config.k3_wgs_per_cu; 0 = this default)
    int wg = 256 * per_cu / (row_tiles < per_cu ? row_tiles : per_cu);
    if (wg < 256) wg = 256;
    const int gx = tiles_total < wg ? tiles_total : wg;
    hipLaunchKernelGGL(kern, dim3(gx, row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
