// 3x3x3 (pad 1) convolution of the 8-channel full-resolution layers (in_block, up5, out_block and their backward-data), bf16 / fp16.
//
// These layers are HBM-bound by the algorithm (8 -> 8 channels at 96^3, B = 2: 57 MB in + out, 7 us at 8 TB/s) and ran
// instruction-issue bound in k3b_kernel<8,...> (31 us): per 64 output voxels a wave spent ~277 VALU, ~129 SALU, 59 LDS and
// 28 MFMA instructions, half of the MFMA rows (8 of 16) and half of the epilogue lanes idle.  This kernel cuts the
// instruction count per voxel ~2.5x:
//   * Toeplitz rows: the 16 MFMA rows are (dx2, co) — two neighbouring output voxels along x times 8 output channels — and a
//     k-group is one (tz, ty) pair with k = (window position xpos 0..3, ci): A[(dx2,co)][(xpos,ci)] = W[co][ci][tz][ty][xpos-dx2]
//     (zero outside 0..2).  One MFMA then yields 32 voxels x 8 channels with every accumulator lane in use, 9 MFMAs per 32
//     voxels instead of 14, and a k-group's B fragment is just the 16 bytes of halo voxel (x = 2*col + xpos): no tap table.
//   * the 9 A fragments (144 B per lane) live in registers for the whole kernel (weights are 8x8x27);
//   * B fragments are shared between the three ty of a (tz, halo row): 30 LDS reads feed the 72 MFMAs of a wave's tile;
//   * 4 x 8 x 32 tiles: halo 2.0x the outputs (4 x 4 x 16: 2.5x), a quarter of the per-voxel tile bookkeeping;
//   * the epilogue's 64 lanes store 512 contiguous bytes per instruction.
// Weights arrive in the Toeplitz fragment order from vs_pack_weight (pack.hip picks it for this shape class, vs_k3_toeplitz).
// Staging, statistics, fused IN-backward sums and the softmax epilogue follow k3b_kernel.
#pragma once
#include <stdlib.h>
#include "igemm.h"

#define K3T_LDS_RED 0          // float[4][8][2]
#define K3T_LDS_TILE 512       // halo tile [6][YT+2][34] x 16 B, then the per-(n,c) tables

template <int YT>
struct K3TGeom {
    static constexpr int PX = 34, PY = YT + 2, PLANE = PX * PY, TV = 6 * PLANE;
    static constexpr int TILE_BYTES = ((TV + 255) / 256) * 256 * 16;     // every thread stores all its NIT fragments (no exec-masked tail)
    static constexpr int NIT = (TV + 255) / 256;
};

// HS: lazy input (normalise + ReLU while staging), compile-time like every condition on the staging path
// T: unsigned short (bf16 bits) or vs_half (fp16); last template argument (kernel-name prefix unchanged)
// FA (backward-data only): the input gradient arrives UN-applied — p.x = g = dL/da of the lazy activation a = relu(norm(p.fa_x)), with that
// activation's statistics (p.x_stats) and IN-backward sums (p.fa_sums) — and the apply pass rstd * (g*[xhat>0] - m1 - xhat * m2) runs while the
// halo tile is staged (the standalone vs_instnorm_relu_bwd_apply launch, 3 tensor passes at 96^3, disappears); centre voxels are also
// written to p.fa_dx when given (the weight gradient of this layer reads the applied gradient).
template <int EPI, bool SUMS, int YT, bool HS, typename T = unsigned short, bool FA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k3t_kernel(const G1Params p) {
    K3_TICK_INIT
    using GEO = K3TGeom<YT>;
    constexpr int PX = GEO::PX, PY = GEO::PY, PLANE = GEO::PLANE, TV = GEO::TV, NIT = GEO::NIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3T_LDS_RED);
    char* s_tile = smem + K3T_LDS_TILE;
    float* s_scale = (float*)(s_tile + GEO::TILE_BYTES);  // rstd and -mean*rstd of the lazy input, [N*8] each
    float* s_shift = s_scale + p.N * 8;
    float* s_mkm = s_shift + p.N * 8;                    // mean / rstd of the mask tensor's channels (fused IN-bwd sums)
    float* s_mkr = s_mkm + p.N * 8;
    float* s_fa = s_mkr + p.N * 8;                       // FA: rstd, -mean*rstd, m1, m2 of the input gradient's activation, [N*8] each
    static_assert(!FA || (!HS && EPI == EPI_RAW), "fused apply: backward-data use (materialised gradient in), with or without the fused IN-backward sums");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int dx2 = g >> 1, c4 = 4 * (g & 1);            // this lane's accumulator rows: output voxel x = 2*col + dx2, channels c4..c4+3
    constexpr bool has_stats = HS;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    const i32x4 frsrc = make_rsrc(FA ? p.fa_x : p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    const i32x4 dxrsrc = make_rsrc(FA && p.fa_dx != nullptr ? p.fa_dx : p.y, (FA && p.fa_dx != nullptr) ? (unsigned int)((long long)p.N * p.D * p.H * p.W * 16) : 0u);

    // the (sum, sumsq) pair this thread turns into a table entry: oldest load in the queue
    const double* st_src = SUMS ? p.mask_stats : p.x_stats;
    const int st_n = SUMS ? p.N * 8 : (has_stats ? p.N * 8 : 0);
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(st_src, (size_t)tid, (size_t)st_n, st_pre);
    double fa_pre[2][2] = {{0.0, 1.0}, {0.0, 0.0}};      // FA: wave 1 requests the activation's (sum, sumsq) and (sum g*mask, sum g*mask*xhat) pairs
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * 8) {
            stat_load(p.x_stats, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[0]);
            stat_load(p.fa_sums, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[1]);
        }
    }

    // ---- per-thread staging geometry: fragment b = halo voxel tid + 256 b -------------------------------------------------
    int rel_off[NIT], tzyx[NIT];
    unsigned int cbits = 0;                              // FA: fragment b is a centre (non-halo) voxel of the tile
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int tv = tid + b * 256;
        const int tx_ = tv % PX, ty_ = (tv / PX) % PY, tz_ = tv / PLANE;
        rel_off[b] = ((tz_ * p.H + ty_) * p.W + tx_) * 16;
        tzyx[b] = tv < TV ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;
        cbits |= (tv < TV && tz_ >= 1 && tz_ <= 4 && ty_ >= 1 && ty_ <= YT && tx_ >= 1 && tx_ <= 32) ? (1u << b) : 0u;
    }
    u32x4 xv[NIT], fv[FA ? NIT : 1];
    unsigned int okbits = 0;
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {
        Coord c;
        c.n = fdiv(t, p.fd_m[0], p.fd_s[0]);
        const int tl = t - c.n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], p.fd_s[1]);
        const int r = tl - tz * (p.txn * p.tyn);
        const int ty = fdiv(r, p.fd_m[2], p.fd_s[2]);
        c.z0 = tz * 4; c.y0 = ty * YT; c.x0 = (r - ty * p.txn) * 32;
        return c;
    };
    auto load_x = [&](const Coord& c) {
        const int base = (((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * 16;
        okbits = 0;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int gz = c.z0 - 1 + (tzyx[b] & 0xff), gy = c.y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = c.x0 - 1 + (tzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            okbits |= ok ? (1u << b) : 0u;
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
            if constexpr (FA) fv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(frsrc, ok ? base + rel_off[b] : -1, 0, 0));
        }
    };
    auto write_x_fa = [&](const Coord& c) {             // FA: apply pass on the staged fragments, [+ the applied gradient of the centre voxels to fa_dx]
        f32x2 r2[4], s2[4], a2[4], b2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r2[i] = *(const f32x2*)(s_fa + 0 * p.N * 8 + c.n * 8 + 2 * i);
            s2[i] = *(const f32x2*)(s_fa + 1 * p.N * 8 + c.n * 8 + 2 * i);
            a2[i] = *(const f32x2*)(s_fa + 2 * p.N * 8 + c.n * 8 + 2 * i);
            b2[i] = *(const f32x2*)(s_fa + 3 * p.N * 8 + c.n * 8 + 2 * i);
        }
        const int base = (((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * 16;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 g2, x2;
                g2[0] = H16<T>::lo(xv[b][i]); g2[1] = H16<T>::hi(xv[b][i]);
                x2[0] = H16<T>::lo(fv[b][i]); x2[1] = H16<T>::hi(fv[b][i]);
                const f32x2 xh = x2 * r2[i] + s2[i];
                f32x2 gm;
                gm[0] = xh[0] > 0.f ? g2[0] : 0.f;
                gm[1] = xh[1] > 0.f ? g2[1] : 0.f;
                const f32x2 d = r2[i] * (gm - a2[i] - xh * b2[i]);
                v[i] = H16<T>::pack2(d);
            }
            const bool ok = (okbits >> b) & 1u;           // out-of-volume halo voxels: the gradient is zero-padded
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] : 0u;
            *(u32x4*)(s_tile + (tid + b * 256) * 16) = v;
            if (p.fa_dx != nullptr)                        // workgroup-uniform
                vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), dxrsrc, (ok && ((cbits >> b) & 1u)) ? base + rel_off[b] : -1, 0, 0);
        }
    };
    auto write_x = [&](int n) {
        f32x2 sc[4], sh[4];
        if (has_stats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = *(const f32x2*)(s_scale + n * 8 + 2 * i);
                sh[i] = *(const f32x2*)(s_shift + n * 8 + 2 * i);
            }
        }
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            u32x4 v = xv[b];
            if (has_stats) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;       // zero padding applies to the normalised activation
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            *(u32x4*)(s_tile + (tid + b * 256) * 16) = v;      // fragments beyond TV are zeros in the padded tail of the tile region
        }
    };

    // ---- first tile in flight, weights into registers, tables -------------------------------------------------------------
    // XCD-aware walk: consecutive workgroup ids land on different XCDs (8, each with its own L2).  XCD x owns the contiguous run
    // [x*T/8, (x+1)*T/8) of the tile list and its workgroups deal that run round-robin, so neighbouring tiles share an L2 AND every
    // XCD gets the same number of tiles (k3b's first walk gave the remainder T mod G to XCD 0 and 1: their CUs ran 9 tiles against
    // 6 elsewhere at 96^3, and the launch took as long as they did).  Identity walk when the grid is not a multiple of 8.
    int t, t_end, G;
    if (((int)gridDim.x & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7;
        G = (int)gridDim.x >> 3;
        t = (int)(((long long)total_tiles * xcd) >> 3) + ((int)blockIdx.x >> 3);
        t_end = (int)(((long long)total_tiles * (xcd + 1)) >> 3);
    } else { G = (int)gridDim.x; t = (int)blockIdx.x; t_end = total_tiles; }
    Coord cur = tile_coord(t < t_end ? t : 0), nxt = cur;
    u32x4 wa[9];
    {
        const u32x4* __restrict__ wp = (const u32x4*)p.wp;
#pragma unroll
        for (int kg = 0; kg < 9; ++kg) wa[kg] = wp[kg * 64 + lane];
    }
    if (t < t_end) load_x(cur);
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (p.bias != nullptr && c4 + r < (EPI == EPI_SOFTMAX2 ? 2 : p.M)) ? p.bias[c4 + r] : 0.f;
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(st_src, (size_t)i, (size_t)st_n, st);
        float m, r;
        stats_to_mean_rstd_fast(st, SUMS ? p.inv_count_out : p.inv_count_in, p.eps, m, r);
        if constexpr (SUMS) { s_mkm[i] = m; s_mkr[i] = r; }
        else { s_scale[i] = r; s_shift[i] = -m * r; }
    }
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * 8) {
            const int i = tid - 64;
            float m, r;
            stats_to_mean_rstd_fast(fa_pre[0], p.inv_count_in, p.eps, m, r);
            s_fa[0 * p.N * 8 + i] = r;
            s_fa[1 * p.N * 8 + i] = -m * r;
            s_fa[2 * p.N * 8 + i] = (float)(fa_pre[1][0] * p.inv_count_in);
            s_fa[3 * p.N * 8 + i] = (float)(fa_pre[1][1] * p.inv_count_in);
        }
    }
    const char* s_b = s_tile + ((wave * PY) * PX + 2 * col + g) * 16;     // B fragment of (tz, halo row yy): + ((tz * PY + yy) * PX) * 16
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[r]));           // the bias wait belongs to the prologue
#pragma unroll
    for (int kg = 0; kg < 9; ++kg) asm volatile("" : "+v"(wa[kg]));
    bool first = true;
    __syncthreads();                                     // tables visible
    K3_TICK(0);

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave, ox = x0 + 2 * col + dx2;
        // byte offset of output voxel (n, oz, y0 + cg, ox), channel c4: ebase + cg * W * 16
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + ox) * 8 + c4) * 2;
        const bool zx_ok = oz < p.D && ox < p.W;
        if (!first) __syncthreads();                     // every wave is done reading the previous tile
        first = false;
        K3_TICK(1);
        if constexpr (FA) write_x_fa(cur); else write_x(n);
        K3_TICK(2);
        __syncthreads();
        K3_TICK(3);
        // requests, oldest-needed first: the mask fragments of this tile, then the next tile's halo
        u32x2 mk[YT];
        if constexpr (SUMS) {
#pragma unroll
            for (int cg = 0; cg < YT; ++cg)
                mk[cg] = __builtin_bit_cast(u32x2, vs_raw_buffer_load_b64(mrsrc, (zx_ok && y0 + cg < p.H) ? ebase + cg * p.W * 16 : -1, 0, 0));
        }
        nxt = tile_coord(t + G < t_end ? t + G : t);
        if (t + G < t_end) load_x(nxt);
        K3_TICK(4);

        // ---- multiply the tile out of LDS: one B fragment per (tz, halo row), used by up to three (ty, output row) pairs ----
        f32x4 acc[YT];
#pragma unroll
        for (int cg = 0; cg < YT; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tz = 0; tz < 3; ++tz) {
            u32x4 fb[PY];
#pragma unroll
            for (int yy = 0; yy < PY; ++yy) fb[yy] = *(const u32x4*)(s_b + ((tz * PY + yy) * PX) * 16);
            // YT independent accumulators per k-group: back-to-back MFMAs never wait on one another (with the halo row outermost,
            // an accumulator came round again after two MFMAs and every MFMA paid its full latency: 29 cycles each, measured)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int cg = 0; cg < YT; ++cg) acc[cg] = mfma16(wa[tz * 3 + ty], fb[cg + ty], acc[cg], (T*)nullptr);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue ----
        K3_TICK(5);
        if constexpr (EPI == EPI_SOFTMAX2) {
            const size_t V = (size_t)p.D * p.H * p.W;
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) {
                const int oy = y0 + cg;
                const bool valid = zx_ok && oy < p.H;
                float l0 = acc[cg][0] + bv[0], l1 = acc[cg][1] + bv[1];
                const size_t v = ((size_t)oz * p.H + oy) * p.W + ox;
                if (p.drop_p > 0.f) {
                    l0 *= dropout_scale(p.drop_seed, ((unsigned long long)n * 2 + 0) * V + v, p.drop_p);
                    l1 *= dropout_scale(p.drop_seed, ((unsigned long long)n * 2 + 1) * V + v, p.drop_p);
                }
                const float mx = fmaxf(l0, l1);
                const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
                const float inv = 1.f / (e0 + e1);
                if (valid && (g & 1) == 0) {
                    p.prob[((size_t)n * 2 + 0) * V + v] = e0 * inv;
                    p.prob[((size_t)n * 2 + 1) * V + v] = e1 * inv;
                }
                if (p.y != nullptr) {
                    // the same probabilities as the next network's channels-last bf16 input (8 stored channels, 2 real): what
                    // vs_pack_planar(prob) would write, without the extra launch and the re-read
                    f32x2 pr;
                    pr[0] = e0 * inv; pr[1] = e1 * inv;
                    i32x2 pk;
                    pk[0] = (g & 1) == 0 ? (int)H16<T>::pack2(pr) : 0;
                    pk[1] = 0;
                    vs_raw_buffer_store_b64(pk, yrsrc, valid ? ebase + cg * p.W * 16 : -1, 0, 0);
                }
            }
        } else {
            float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SUMS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * 8 + c4 + r]; mr[r] = s_mkr[n * 8 + c4 + r]; }
            }
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) {
                const bool valid = zx_ok && y0 + cg < p.H;
                // round once to T; the statistics are those of the stored values
                f32x2 lo, hi;
                lo[0] = acc[cg][0] + bv[0]; lo[1] = acc[cg][1] + bv[1];
                hi[0] = acc[cg][2] + bv[2]; hi[1] = acc[cg][3] + bv[3];
                i32x2 pk;
                pk[0] = (int)H16<T>::pack2(lo);
                pk[1] = (int)H16<T>::pack2(hi);
                vs_raw_buffer_store_b64(pk, yrsrc, valid ? ebase + cg * p.W * 16 : -1, 0, 0);
                float v[4];
                v[0] = H16<T>::lo((unsigned int)pk[0]); v[1] = H16<T>::hi((unsigned int)pk[0]);
                v[2] = H16<T>::lo((unsigned int)pk[1]); v[3] = H16<T>::hi((unsigned int)pk[1]);
                if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                if constexpr (SUMS) {
                    const u32x2 xx = mk[cg];
                    float xv4[4];
                    xv4[0] = H16<T>::lo(xx[0]); xv4[1] = H16<T>::hi(xx[0]);
                    xv4[2] = H16<T>::lo(xx[1]); xv4[3] = H16<T>::hi(xx[1]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (xv4[r] - mm[r]) * mr[r];
                        const float gm = xh > 0.f ? v[r] : 0.f;
                        ssum[r] += gm; ssq[r] += gm * xh;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[r] += v[r]; ssq[r] += v[r] * v[r]; }
                }
            }
            double* const red_dst0 = SUMS ? p.sums : p.y_stats;
            double* const red_dst = red_dst0;
            if (red_dst != nullptr) {
                const bool flush = t + G >= t_end || nxt.n != n;       // workgroup-uniform
                if (flush) {
                    // lanes of equal (g & 1) hold the same 4 channels: fold the 16 columns and the two dx2 halves
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = ssum[r], q = ssq[r];
                        { s = row16_sum(s); q = row16_sum(q); }      // DPP: the four-step __shfl_xor butterfly was four ds_bpermute round trips per statistic
                        s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                        if (col == 0 && dx2 == 0) {
                            s_red[(wave * 8 + c4 + r) * 2 + 0] = s;
                            s_red[(wave * 8 + c4 + r) * 2 + 1] = q;
                        }
                        ssum[r] = 0.f; ssq[r] = 0.f;
                    }
                    __syncthreads();
                    if (tid < 16) {
                        const int ch = tid >> 1, st = tid & 1;
                        if (ch < p.M) {
                            const double tot = (double)s_red[(0 * 8 + ch) * 2 + st] + (double)s_red[(1 * 8 + ch) * 2 + st] +
                                               (double)s_red[(2 * 8 + ch) * 2 + st] + (double)s_red[(3 * 8 + ch) * 2 + st];
                            stat_add(red_dst, (size_t)n * 8 + ch, (size_t)p.N * 8, st, tot);
                        }
                    }
                    // s_red is rewritten only after the two barriers at the top of the next tile
                }
            }
        }
        cur = nxt;
        K3_TICK(6);
    }
    K3_TICK_FLUSH;
}

template <typename T, int EPI, bool SUMS, int YT, bool HS, bool FA = false>
static int k3t_launch_t(const G1Params& p_in, hipStream_t stream) {
    using GEO = K3TGeom<YT>;
    G1Params p = p_in;
    if (p.C != 8 || p.M != 8) return VS_ESHAPE;
    if (FA && p.N * 8 > 192) return VS_ESHAPE;          // wave 1 .. 3 build the fused-apply tables
    const size_t lds = K3T_LDS_TILE + (size_t)GEO::TILE_BYTES + (size_t)(FA ? 8 : 4) * p.N * 8 * sizeof(float);
    if (lds > 160 * 1024) return VS_ESHAPE;
    p.txn = (p.W + 31) / 32;
    p.tyn = (p.H + YT - 1) / YT;
    p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
    const long long tiles = (long long)p.tiles_per_sample * p.N;
    // buffer offsets are 32-bit bytes, signed on the device
    if ((long long)p.N * p.D * p.H * p.W * 16 >= 2147483648ll || tiles >= 2147483647ll) return VS_ESHAPE;
    k3b_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k3b_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k3b_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    if (SUMS != (p.sums != nullptr) || (SUMS && !FA && p.x_stats != nullptr)) return VS_EINVAL;
    if (FA && (!p.x_stats || !p.fa_x || !p.fa_sums)) return VS_EINVAL;
    auto kern = k3t_kernel<EPI, SUMS, YT, HS, T, FA>;
    static const hipError_t attr_err =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    // persistent grid: two workgroups per CU are resident (185-218 VGPRs).  Measured at 96^3, B = 2 (57 MB algorithmic, 62 MB of HBM
    // traffic by the PMC counters): 21-24 us per launch whichever of 2 / 3 workgroups per CU, 4x4x32 or 4x8x32 tiles, or MFMA loop order
    // is used — ~5 us of that is the launch itself, the rest moves ~4 TB/s (the pure streaming kernels of this library reach 4.7).
    const int per_cu = vs_cfg().k3t_wgs_per_cu;
    const int cap = 256 * per_cu;
    const int gx = tiles < cap ? (int)tiles : cap;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T, int EPI, bool SUMS, int YT>
static int k3t_launch(const G1Params& p, hipStream_t stream) {
    if constexpr (EPI == EPI_RAW) {
        if (p.fa_x != nullptr) return k3t_launch_t<T, EPI, SUMS, YT, false, true>(p, stream);
    }
    if (!SUMS && p.x_stats != nullptr) return k3t_launch_t<T, EPI, SUMS, YT, !SUMS>(p, stream);
    return k3t_launch_t<T, EPI, SUMS, YT, false>(p, stream);
}
