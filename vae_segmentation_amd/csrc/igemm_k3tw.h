// Backward-data of the 8 -> 8 3x3x3 layers at full resolution WITH THE LAYER'S WEIGHT GRADIENT (round 5), bf16 / fp16.
//
// k3tw_kernel = k3t_kernel<EPI_RAW, SUMS, 8, false, T, FA> (igemm_k3t.h: backward-data with the fused IN-backward sums [and the fused apply of its input
// gradient]) plus the weight gradient of the same layer from the operands that launch already holds:
//   * P, the applied output gradient, is the halo tile the backward-data GEMM multiplies out of LDS ([6][10][34] voxels x 8 channels);
//   * Q, the layer's input activation relu(norm(x)), is formed in the epilogue anyway (the fused sums need xhat of the raw input under every output voxel):
//     it goes to a second LDS tile ([4][8][32] voxels x 8 channels, no halo).
// The grouped weight-gradient launch (wgrad.hip) reads both tensors again from HBM (and, after a fused apply, a stored copy of the applied gradient that exists
// for it alone); here neither is read twice and that copy is never written.
// Operands exchanged as in wgrad.hip multi_plan:  S[o][c][m] = sum_v Q(v)[c] * P(v + o)[m]  = dW[m][c][-o], K = voxels through ds_read_b64_tr_b16.
// One MFMA (16x16x32) covers a whole 32-voxel x row: rows (r, c) = the two y rows of a pair x 8 input channels, columns (t, m) = two neighbouring x taps x 8
// gradient channels, against halo row 2s + dy' (dy' = 0..3): row r meets tap dy = dy' - r.  24 blocks (dz 3 x dy' 4 x dx pair 2: 27 useful taps of 48
// computed) x 16 K-steps per tile = 384 MFMAs and 184 transposing read pairs per tile next to the backward-data's 288 MFMAs; the four waves split the BLOCKS (six each: 24 accumulator registers,
// parked in LDS between tiles — the kernel stays at two waves per SIMD) and walk all K-steps.  One slab [27][8][8] per workgroup; the grouped reduction
// (g3_reduce_group_kernel, VS_WGRAD_SLABS descriptors) sums them in a fixed order.
// Measured standalone at 2 x 96^3 (tools/k3tw_probe.py, profiles/r05_k3tw_probe.txt): +4.7 us on the fused-apply launch (which no longer stores the applied
// gradient), +13.8 us on the plain one, against 22.5 us per layer in the grouped launch; in the step: profiles/r05_ab_fuse_wgrad_*.json.
#pragma once
#include <stdlib.h>
#include "igemm.h"
#include "igemm_k3t.h"

typedef __attribute__((ext_vector_type(4))) short k3tw_s16x4;
typedef __attribute__((address_space(3))) k3tw_s16x4 k3tw_lds_s16x4;
__device__ __forceinline__ u32x4 pw_tr_pair(const char* s_base, int off0, int off1) {       // eight 16-bit k-values of one row / column (wgrad.hip tr_pair)
    const k3tw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((k3tw_lds_s16x4*)(s_base + off0));
    const k3tw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((k3tw_lds_s16x4*)(s_base + off1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(u32x4, v);
}

#define K3TW_Q_BYTES 16384                 // [4][8][32] voxels x 16 B
#define K3TW_ACC_BYTES (4 * 6 * 64 * 16)   // [wave][block][lane] f32x4
#define K3TW_SLAB_ELEMS 1728               // [tap 27][c 8][m 8]

// MODE: how the gradient tile is staged — 0: the applied gradient as stored; 1 (FA): un-applied, with the IN-backward apply of its activation (k3t_kernel FA);
// 2 (SM, out_block): the two-class softmax backward of planar probabilities and their gradients [+ a channels-last gradient part], logits' dropout included —
// the launch vs_softmax2_cl_bwd and the tensor it wrote disappear.  WG: form the weight gradient (the frozen VAE's out_block wants SM without it).
template <typename T, int MODE, bool WG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k3tw_kernel(const G1Params p) {
    constexpr int YT = 8;
    constexpr bool FA = MODE == 1, SM = MODE == 2;
    static_assert(WG || SM, "without the weight gradient this is k3t_kernel, except for the softmax-backward staging");
    using GEO = K3TGeom<YT>;
    constexpr int PX = GEO::PX, PY = GEO::PY, PLANE = GEO::PLANE, TV = GEO::TV, NIT = GEO::NIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3T_LDS_RED);
    char* s_tile = smem + K3T_LDS_TILE;
    char* s_q = s_tile + GEO::TILE_BYTES;
    f32x4* s_acc = (f32x4*)(s_q + K3TW_Q_BYTES);
    float* s_scale = (float*)((char*)s_acc + K3TW_ACC_BYTES);
    float* s_shift = s_scale + p.N * 8;
    float* s_mkm = s_shift + p.N * 8;
    float* s_mkr = s_mkm + p.N * 8;
    float* s_fa = s_mkr + p.N * 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int dx2 = g >> 1, c4 = 4 * (g & 1);
    const int total_tiles = p.tiles_per_sample * p.N;
    const long long vol = (long long)p.D * p.H * p.W;
    const i32x4 xrsrc = make_rsrc(SM ? (const void*)p.sm_prob : p.x, (unsigned int)(SM ? p.N * vol * 8 : p.N * vol * 16));
    const i32x4 frsrc = make_rsrc(FA ? p.fa_x : (SM ? p.sm_gcl : p.x), (unsigned int)((FA || (SM && p.sm_gcl != nullptr)) ? p.N * vol * 16 : 0));
    const i32x4 grsrc = make_rsrc(SM ? (const void*)p.sm_gprob : p.x, (unsigned int)((SM && p.sm_gprob != nullptr) ? p.N * vol * 8 : 0));      // absent parts read as zero

    const double* st_src = p.mask_stats;
    const int st_n = p.N * 8;
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(st_src, (size_t)tid, (size_t)st_n, st_pre);
    double fa_pre[2][2] = {{0.0, 1.0}, {0.0, 0.0}};
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * 8) {
            stat_load(p.x_stats, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[0]);
            stat_load(p.fa_sums, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[1]);
        }
    }

    int rel_off[NIT], tzyx[NIT];
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int tv = tid + b * 256;
        const int tx_ = tv % PX, ty_ = (tv / PX) % PY, tz_ = tv / PLANE;
        rel_off[b] = ((tz_ * p.H + ty_) * p.W + tx_) * 16;
        tzyx[b] = tv < TV ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;
    }
    u32x4 xv[NIT], fv[(FA || SM) ? NIT : 1];
    unsigned int okbits = 0;
    unsigned int cbits = 0;                              // SM: fragment b is a centre (non-halo) voxel of the tile (the bias gradient sums those)
    float bsum[2] = {0.f, 0.f};
    if constexpr (SM) {
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int tv = tid + b * 256;
            const int tx_ = tv % PX, ty_ = (tv / PX) % PY, tz_ = tv / PLANE;
            cbits |= (tv < TV && tz_ >= 1 && tz_ <= 4 && ty_ >= 1 && ty_ <= YT && tx_ >= 1 && tx_ <= 32) ? (1u << b) : 0u;
        }
    }
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
            if constexpr (SM) {
                // planar fp32 [N][2][V]: voxel index within the sample = (base + rel_off) / 16 - n V
                const int vi = ((base + rel_off[b]) >> 4) - c.n * (int)vol;
                const int o0 = ok ? ((c.n * 2) * (int)vol + vi) * 4 : -1, o1 = ok ? ((c.n * 2 + 1) * (int)vol + vi) * 4 : -1;
                xv[b][0] = (unsigned int)vs_raw_buffer_load_b32(xrsrc, o0, 0, 0);
                xv[b][1] = (unsigned int)vs_raw_buffer_load_b32(xrsrc, o1, 0, 0);
                xv[b][2] = (unsigned int)vs_raw_buffer_load_b32(grsrc, o0, 0, 0);
                xv[b][3] = (unsigned int)vs_raw_buffer_load_b32(grsrc, o1, 0, 0);
                fv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(frsrc, ok ? base + rel_off[b] : -1, 0, 0));
            } else {
                xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
                if constexpr (FA) fv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(frsrc, ok ? base + rel_off[b] : -1, 0, 0));
            }
        }
    };
    auto write_x_sm = [&](const Coord& c) {              // softmax2_bwd_kernel (misc.hip) on the staged voxels
        const int base = (((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * 16;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const float p0 = __uint_as_float(xv[b][0]), p1 = __uint_as_float(xv[b][1]);
            const float g0 = __uint_as_float(xv[b][2]) + H16<T>::lo(fv[b][0]), g1 = __uint_as_float(xv[b][3]) + H16<T>::hi(fv[b][0]);
            const float dot = p0 * g0 + p1 * g1;
            f32x2 f;
            f[0] = p0 * (g0 - dot);
            f[1] = p1 * (g1 - dot);
            if (p.drop_p > 0.f) {                         // workgroup-uniform: backward of the logit dropout fused into the out_block epilogue
                const long long vi = (long long)((base + rel_off[b]) >> 4) - (long long)c.n * vol;
                f[0] *= dropout_scale(p.drop_seed, ((unsigned long long)c.n * 2 + 0) * (unsigned long long)vol + (unsigned long long)vi, p.drop_p);
                f[1] *= dropout_scale(p.drop_seed, ((unsigned long long)c.n * 2 + 1) * (unsigned long long)vol + (unsigned long long)vi, p.drop_p);
            }
            const bool ok = (okbits >> b) & 1u;
            u32x4 v = u32x4{ok ? H16<T>::pack2(f) : 0u, 0u, 0u, 0u};
            *(u32x4*)(s_tile + (tid + b * 256) * 16) = v;
            if (ok && ((cbits >> b) & 1u)) { bsum[0] += H16<T>::lo(v[0]); bsum[1] += H16<T>::hi(v[0]); }      // the stored (rounded) values, as vs_bias_grad summed them
        }
    };
    auto write_x_fa = [&](const Coord& c) {
        f32x2 r2[4], s2[4], a2[4], b2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r2[i] = *(const f32x2*)(s_fa + 0 * p.N * 8 + c.n * 8 + 2 * i);
            s2[i] = *(const f32x2*)(s_fa + 1 * p.N * 8 + c.n * 8 + 2 * i);
            a2[i] = *(const f32x2*)(s_fa + 2 * p.N * 8 + c.n * 8 + 2 * i);
            b2[i] = *(const f32x2*)(s_fa + 3 * p.N * 8 + c.n * 8 + 2 * i);
        }
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
            const bool ok = (okbits >> b) & 1u;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] : 0u;
            *(u32x4*)(s_tile + (tid + b * 256) * 16) = v;
            // the applied gradient is not stored: its only reader was the weight gradient, which runs here
        }
    };
    auto write_x = [&](int n) {
#pragma unroll
        for (int b = 0; b < NIT; ++b) *(u32x4*)(s_tile + (tid + b * 256) * 16) = xv[b];
    };

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
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 16));
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(st_src, (size_t)i, (size_t)st_n, st);
        float m, r;
        stats_to_mean_rstd_fast(st, p.inv_count_out, p.eps, m, r);
        s_mkm[i] = m; s_mkr[i] = r;
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
    const char* s_b = s_tile + ((wave * PY) * PX + 2 * col + g) * 16;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < 9; ++kg) asm volatile("" : "+v"(wa[kg]));

    // ---- weight-gradient geometry: this wave's six blocks, this lane's transposing-read addresses ----
    const int q4 = col >> 2, p4 = col & 3, hi2 = p4 >> 1, half = p4 & 1;
    // A (Q tile): lane reads 8 bytes = channels 4*half .. of voxel (z, 2s + hi2, 16h + 4g + q4)
    const int a_lane = ((hi2 * 32) + 4 * g + q4) * 16 + half * 8;
    // B (P halo tile): channels 4*half .. of voxel (z + dz, 2s + dy', 16h + 4g + q4 + dx), dx = 2 dxb + hi2 (dxb = 1: dx = 2 for both column halves)
    const int b_lane = (4 * g + q4) * 16 + half * 8;
    int b_blk[6];
    if constexpr (WG) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int blk = wave + 4 * j;
            const int dz = blk >> 3, dyp = (blk & 7) >> 1, dxb = blk & 1;
            b_blk[j] = ((dz * PY + dyp) * PX + (dxb ? 2 : hi2)) * 16 + b_lane;
            s_acc[(wave * 6 + j) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};       // own slot: read back by this lane only until the end
        }
    }
    bool first = true;
    __syncthreads();

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave, ox = x0 + 2 * col + dx2;
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + ox) * 8 + c4) * 2;
        const bool zx_ok = oz < p.D && ox < p.W;
        if (!first) __syncthreads();
        first = false;
        if constexpr (FA) write_x_fa(cur); else if constexpr (SM) write_x_sm(cur); else write_x(n);
        __syncthreads();
        u32x2 mk[YT];
#pragma unroll
        for (int cg = 0; cg < YT; ++cg)
            mk[cg] = __builtin_bit_cast(u32x2, vs_raw_buffer_load_b64(mrsrc, (zx_ok && y0 + cg < p.H) ? ebase + cg * p.W * 16 : -1, 0, 0));
        nxt = tile_coord(t + G < t_end ? t + G : t);
        if (t + G < t_end) load_x(nxt);

        f32x4 acc[YT];
#pragma unroll
        for (int cg = 0; cg < YT; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tz = 0; tz < 3; ++tz) {
            u32x4 fb[PY];
#pragma unroll
            for (int yy = 0; yy < PY; ++yy) fb[yy] = *(const u32x4*)(s_b + ((tz * PY + yy) * PX) * 16);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int cg = 0; cg < YT; ++cg) acc[cg] = mfma16(wa[tz * 3 + ty], fb[cg + ty], acc[cg], (T*)nullptr);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue (k3t_kernel's) + the Q tile ----
        {
            float mm[4], mr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * 8 + c4 + r]; mr[r] = s_mkr[n * 8 + c4 + r]; }
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) {
                const bool valid = zx_ok && y0 + cg < p.H;
                f32x2 lo, hi;
                lo[0] = acc[cg][0]; lo[1] = acc[cg][1];
                hi[0] = acc[cg][2]; hi[1] = acc[cg][3];
                i32x2 pk;
                pk[0] = (int)H16<T>::pack2(lo);
                pk[1] = (int)H16<T>::pack2(hi);
                vs_raw_buffer_store_b64(pk, yrsrc, valid ? ebase + cg * p.W * 16 : -1, 0, 0);
                float v[4];
                v[0] = H16<T>::lo((unsigned int)pk[0]); v[1] = H16<T>::hi((unsigned int)pk[0]);
                v[2] = H16<T>::lo((unsigned int)pk[1]); v[3] = H16<T>::hi((unsigned int)pk[1]);
                if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                const u32x2 xx = mk[cg];
                float xv4[4], qv[4];
                xv4[0] = H16<T>::lo(xx[0]); xv4[1] = H16<T>::hi(xx[0]);
                xv4[2] = H16<T>::lo(xx[1]); xv4[3] = H16<T>::hi(xx[1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (xv4[r] - mm[r]) * mr[r];
                    const float gm = xh > 0.f ? v[r] : 0.f;
                    ssum[r] += gm; ssq[r] += gm * xh;
                    // Q as the forward conv consumed it (normalise-on-load: x * rstd - mean * rstd, one fma, then the ReLU): bit-identical to the operand the
                    // grouped weight-gradient launch forms with act8
                    qv[r] = valid ? fmaxf(fmaf(xv4[r], mr[r], -mm[r] * mr[r]), 0.f) : 0.f;
                }
                if constexpr (WG) {
                    f32x2 q0, q1;
                    q0[0] = qv[0]; q0[1] = qv[1]; q1[0] = qv[2]; q1[1] = qv[3];
                    *(u32x2*)(s_q + ((wave * 8 + cg) * 32 + 2 * col + dx2) * 16 + c4 * 2) = u32x2{H16<T>::pack2(q0), H16<T>::pack2(q1)};
                }
            }
            const bool flush = t + G >= t_end || nxt.n != n;
            if (flush) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = ssum[r], q = ssq[r];
                    { s = row16_sum(s); q = row16_sum(q); }
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
                    const double tot = (double)s_red[(0 * 8 + ch) * 2 + st] + (double)s_red[(1 * 8 + ch) * 2 + st] +
                                       (double)s_red[(2 * 8 + ch) * 2 + st] + (double)s_red[(3 * 8 + ch) * 2 + st];
                    stat_add(p.sums, (size_t)n * 8 + ch, (size_t)p.N * 8, st, tot);
                }
            }
        }

        // ---- weight gradient of this tile: every wave walks the 16 K-steps (z slice, y pair) for its six blocks ----
        if constexpr (WG) {
            __syncthreads();                             // the Q tile of all four z slices is complete
            f32x4 wacc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) wacc[j] = s_acc[(wave * 6 + j) * 64 + lane];
            // this wave's blocks are (dz = j >> 1, dy' = a + 2 (j & 1), dx pair w & 1), a = w >> 1.  A halo-row fragment depends on the tile plane tz = z + dz, the
            // row and the dx pair only: plane tz serves up to three (z, dz) combinations, and within a combination row a + 2 s + 2 serves (s, dy' = a + 2) and
            // (s + 1, dy' = a).  Walking the six planes, five transposing read pairs feed up to 24 MFMAs (one pair per MFMA in the first version, then five per
            // eight): 46 read pairs per tile and wave instead of 112 / 76 — these reads are what the phase costs.
            u32x4 af[4][4];
#pragma unroll
            for (int z = 0; z < 4; ++z)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ao = a_lane + ((z * 8 + 2 * s) * 32) * 16;
                    af[z][s] = pw_tr_pair(s_q, ao, ao + 16 * 16);
                }
            const int b_row0 = b_blk[0];                 // block (dz = 0, dy' = a): its plane-0 address; plane tz and row step are immediates
#pragma unroll
            for (int tz = 0; tz < 6; ++tz) {
                u32x4 rows[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int bn = b_row0 + ((tz * PY + 2 * k) * PX) * 16;
                    rows[k] = pw_tr_pair(s_tile, bn, bn + 16 * 16);
                }
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int z = tz - dz;
                    if (z >= 0 && z <= 3) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            wacc[2 * dz] = mfma16(af[z][s], rows[s], wacc[2 * dz], (T*)nullptr);
                            wacc[2 * dz + 1] = mfma16(af[z][s], rows[s + 1], wacc[2 * dz + 1], (T*)nullptr);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) s_acc[(wave * 6 + j) * 64 + lane] = wacc[j];
        }
        cur = nxt;
    }

    // ---- this workgroup's slab [tap 27][c 8][m 8]: a tap's sum arrives in two parts (the even y rows in block dy' = dy, the odd ones in block dy' = dy + 1,
    // held by different waves); they meet here and the slab goes out as contiguous floats ----
    __syncthreads();
    if constexpr (SM) {
        if (p.wg_bias != nullptr) {                      // this workgroup's share of the bias gradient: sum of the staged centre voxels' logit gradients
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float sb = bsum[k];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sb += __shfl_xor(sb, off, 64);
                if (lane == 0) s_red[wave * 2 + k] = sb;
            }
            __syncthreads();
            if (tid < 2) p.wg_bias[(size_t)blockIdx.x * 2 + tid] = (double)s_red[tid] + (double)s_red[2 + tid] + (double)s_red[4 + tid] + (double)s_red[6 + tid];
        }
    }
    if constexpr (WG) {
        float* wsg = p.wg_ws + (size_t)blockIdx.x * 1728;
        const float* s_accf = (const float*)s_acc;
        for (int o = tid; o < 1728; o += 256) {
            const int tap = o >> 6, c = (o >> 3) & 7, m = o & 7;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int dxb = dx >> 1, colx = 8 * (dx & 1) + m;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int blk = dz * 8 + (dy + r) * 2 + dxb;
                const int ln = 16 * (2 * r + (c >> 2)) + colx;
                v += s_accf[((((blk & 3) * 6 + (blk >> 2)) * 64 + ln) << 2) + (c & 3)];
            }
            wsg[o] = v;
        }
    }
}

// workgroups (= slabs) of the launch for this volume: the persistent grid of k3t_launch_t
static inline int k3tw_grid(int n, int d, int h, int w) {
    const int per_cu = vs_cfg().k3t_wgs_per_cu;
    const long long tiles = (long long)((d + 3) / 4) * ((h + 7) / 8) * ((w + 31) / 32) * n;
    const int cap = 256 * per_cu;
    return tiles < cap ? (int)tiles : cap;
}

template <typename T>
static int k3tw_launch(const G1Params& p_in, hipStream_t stream) {
    using GEO = K3TGeom<8>;
    G1Params p = p_in;
    if (p.C != 8 || p.M != 8 || p.N * 8 > 192) return VS_ESHAPE;
    if (!p.sums || !p.mask_x || !p.mask_stats) return VS_EINVAL;
    const bool fa = p.fa_x != nullptr, sm = p.sm_prob != nullptr, wg = p.wg_ws != nullptr;
    if (fa && (sm || !p.x_stats || !p.fa_sums)) return VS_EINVAL;
    if (sm ? (!p.sm_gprob && !p.sm_gcl) : (!wg || !p.x)) return VS_EINVAL;
    if ((long long)p.N * p.D * p.H * p.W * 8 >= 2147483648ll) return VS_ESHAPE;       // planar fp32 byte offsets are 32-bit too
    const size_t lds = K3T_LDS_TILE + (size_t)GEO::TILE_BYTES + K3TW_Q_BYTES + K3TW_ACC_BYTES + (size_t)8 * p.N * 8 * sizeof(float);
    p.txn = (p.W + 31) / 32;
    p.tyn = (p.H + 7) / 8;
    p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
    const long long tiles = (long long)p.tiles_per_sample * p.N;
    if ((long long)p.N * p.D * p.H * p.W * 16 >= 2147483648ll || tiles >= 2147483647ll) return VS_ESHAPE;
    k3b_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k3b_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k3b_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    const int gx = k3tw_grid(p.N, p.D, p.H, p.W);
#define K3TW_GO(MODEV, WGV)                                                                                                              \
    {                                                                                                                                    \
        auto kern = k3tw_kernel<T, MODEV, WGV>;                                                                                          \
        static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (attr_err != hipSuccess) return (int)attr_err;                                                                                \
        hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, stream, p);                                                                   \
    }
    if (sm) { if (wg) K3TW_GO(2, true) else K3TW_GO(2, false) }
    else if (fa) K3TW_GO(1, true)
    else K3TW_GO(0, true)
#undef K3TW_GO
    VS_CHECK_LAUNCH();
    return VS_OK;
}
