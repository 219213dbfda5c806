// The first two layers of an `Up` block as ONE convolution (bf16 / fp16): ConvTranspose3d(C, C, 2, stride 2) followed — with nothing in
// between (joint_model.py:116-120) — by Conv3d(C, Co, 3, padding 1).
//
// Both are linear, so their composition is one operator on the COARSE grid.  For a fine output voxel f = 2v + p (p = parity in {0,1}^3)
//     y[2v + p][co] = sum over o in O(p) of  [v + o inside the coarse volume] * ( sum_ci Weff[p][o][co][ci] * a[v + o][ci]  +  beta[p][o][co] )
// with O(0) = {-1, 0}, O(1) = {0, +1} per axis (8 coarse neighbours per parity), a = relu(instnorm(x)) the block's (lazy) input,
//     Weff[p][o][co][ci] = sum over (d, t) in S(p, o) of sum_cm W3[co][cm][d] * W2[ci][cm][t],
// S(p, o) per axis: p=0,o=-1 -> {(d=-1,t=1)}; p=0,o=0 -> {(0,0),(+1,1)}; p=1,o=0 -> {(-1,0),(0,1)}; p=1,o=+1 -> {(+1,0)}
// (d = tap of the 3x3x3 conv, t = tap of the transposed conv), and beta the transposed conv's bias seen through the 3x3x3 taps that stay
// inside the FINE volume (zero padding of the intermediate tensor: a constant per output channel in the interior, different on the faces —
// the kernel adds a [27 boundary classes][Co] table).  64 (p, o) blocks of Co x Cin instead of 27 x Co x C + 8 x C x C per coarse voxel:
// 3.4x fewer multiply-adds, no intermediate tensor (113 MB written and re-read at 96^3 x 16), one launch instead of two — in the forward,
// in the backward-data direction (a 4x4x4-tap stride-2 gather: k4g_kernel) and for the weight gradient.
//
//   k4t_kernel  forward:  GEMM rows = (p, co), columns = coarse voxels, k = (o, ci); the 6x6x18 coarse halo tile is staged as in k3b_kernel
//               (bounds-checked buffer loads one stage ahead, InstanceNorm+ReLU on load, XOR-swizzled 32-channel tile); every 16-row block has
//               its own list of 8 neighbour taps (12 for Co = 8, where a row block holds both x parities); the epilogue scatters row (p, co)
//               of column v to fine voxel 2v + p, adds the bias table and accumulates the (sum, sumsq) statistics of the stored values.
//   k4g_kernel  backward-data: rows = ci, columns = coarse voxels, k = (delta = v' - v, p, co) over the fine gradient read SPACE-TO-DEPTH
//               (fine voxel 2v' + p = "channel group p" of coarse voxel v'); every 32-channel chunk has its own tap list; epilogue of k3b_kernel
//               (store + fused InstanceNorm-backward sums of the block's lazy input).
#pragma once
#include <stdlib.h>
#include "igemm.h"

#ifdef VS_STAMPS   // diagnostic build only (tools/build_stamps.sh, tools/stamps_k4.py): per-phase cycle sums of wave 0
__device__ unsigned long long g_k4_stamps[2048 * 8];
extern "C" int vs_debug_read_k4_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k4_stamps), sizeof(unsigned long long) * n);
}
#define K4_TICK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk_acc[i] += now_ - tk_last; tk_last = now_; } while (0)
#define K4_TICK_INIT unsigned long long tk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk_last = __builtin_amdgcn_s_memtime(); tk_acc[7] = __builtin_amdgcn_s_memrealtime();
#define K4_TICK_FLUSH do { if (threadIdx.x == 0) { tk_acc[7] = (tk_acc[7] << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull); for (int i_ = 0; i_ < 8; ++i_) g_k4_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 2047) * 8 + i_] = tk_acc[i_]; } } while (0)
#else
#define K4_TICK(i)
#define K4_TICK_INIT
#define K4_TICK_FLUSH
#endif

#define K4_LDS_RED 0           // float[4][64][2] + float[64][2]
#define K4_LDS_TAPS 2560       // int[<= 512]: tap codes / per-lane-group tap offsets
#define K4_LDS_TILE 4608       // halo tile, weight block, tables

// tap code of a coarse neighbour (dz, dy, dx in 0..2, i.e. offset - 1 .. +1) = dz * 9 + dy * 3 + dx; 27 = none (zero weights, reads the centre)
struct K4Geom {
    static constexpr int TV = 648, PLANE = 108;              // 6 x 6 x 18 halo of a 4 x 4 x 16 tile
};

// ---------------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------------
// CK: input-channel chunk (16: Cin = 16, Co = 8, 12 taps = 6 k-groups per row block; 32: 8 taps = 8 k-groups per row block and chunk)
// RB: 16-row blocks per workgroup.  Row blocks are numbered (p, co / 16) [Co >= 16] or (pz, py) with rows (px, co) [Co = 8].
// W1: single input-channel chunk (Cin <= 32) — the weight block is staged once, not per tile
template <int CK, int RB, bool HS, typename T, bool W1 = (CK == 16)>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CK == 32 && RB == 4 ? 1 : (CK == 16 && RB == 2 ? 3 : 2), CK == 16 && RB == 2 ? 4 : 2))) void k4t_kernel(const G1Params p) {
    K4_TICK_INIT
    constexpr int TV = K4Geom::TV, PLANE = K4Geom::PLANE;
    constexpr int NT = CK == 32 ? 8 : 6;                 // k-groups per (row block, chunk)
    constexpr int CKB = CK * 2, U = CKB / 16, NU = TV * U, NIT = (NU + 255) / 256;
    constexpr int NWF = RB * NT * 64, NWI = (NWF + 255) / 256;
    constexpr int TILE_BYTES = ((NU + 255) / 256) * 256 * 16, W_BYTES = ((NWF + 255) / 256) * 256 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K4_LDS_RED);
    int* s_taps = (int*)(smem + K4_LDS_TAPS);
    char* s_tile = smem + K4_LDS_TILE;
    char* s_w = s_tile + TILE_BYTES;
    float* s_scale = (float*)(s_w + W_BYTES);
    float* s_shift = s_scale + p.N * p.C;
    float* s_bt = s_shift + p.N * p.C;                   // bias table of this workgroup's rows: [27 boundary classes][RB * 16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    const int Co = p.up_co, nb = Co >= 16 ? Co / 16 : 1;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 2));
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    double st_pre[2] = {0.0, 1.0};
    const int st_n = HS ? p.N * p.C : 0;
    if (tid < st_n) stat_load(p.x_stats, (size_t)tid, (size_t)st_n, st_pre);

    const int part = tid % U;
    int rel_off[NIT], tzyx[NIT];
    const int lds_w0 = (tid / U) * CKB;
    unsigned int swzbits = 0;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u / U;
        const int tx_ = tv % 18, ty_ = (tv / 18) % 6, tz_ = tv / PLANE;
        rel_off[b] = (((tz_ * p.H + ty_) * p.W + tx_) * p.C + part * 8) * 2;
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;
        if (CK == 32) swzbits |= (unsigned int)((tx_ >> 2) & 1) << b;
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int f = tid + i * 256;
        if (f > NWF - 1) f = NWF - 1;
        const int rb = f / (NT * 64), r = f - rb * (NT * 64);
        w_off[i] = (rb0 + rb) * (p.nch * NT * 64) + r;                                     // + ch * NT * 64
    }
    u32x4 xv[NIT], wv[NWI];
    unsigned int okbits = 0;
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {
        Coord c;
        c.n = fdiv(t, p.fd_m[0], p.fd_s[0]);
        const int tl = t - c.n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], p.fd_s[1]);
        const int r = tl - tz * (p.txn * p.tyn);
        const int ty = fdiv(r, p.fd_m[2], p.fd_s[2]);
        c.z0 = tz * 4; c.y0 = ty * 4; c.x0 = (r - ty * p.txn) * 16;
        return c;
    };
    auto load_w = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + ch * (NT * 64)];
    };
    auto load_x = [&](const Coord& c, int ch) {
        const int base = ((((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * p.C + ch * CK) * 2;
        okbits = 0;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int gz = c.z0 - 1 + (tzyx[b] & 0xff), gy = c.y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = c.x0 - 1 + (tzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            okbits |= ok ? (1u << b) : 0u;
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
        }
    };
    auto write_x = [&](int n, int ch) {
        f32x2 sc[4], sh[4];
        if (HS) {
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
            if (HS) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            const int pw = CK == 32 ? (part ^ (int)(((swzbits >> b) & 1u) << 1)) : part;
            *(u32x4*)(s_tile + lds_w0 + b * 4096 + pw * 16) = v;
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i) *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    };

    int t, t_end, G;
    if (((int)gridDim.x & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7;
        G = (int)gridDim.x >> 3;
        t = (int)(((long long)total_tiles * xcd) >> 3) + ((int)blockIdx.x >> 3);
        t_end = (int)(((long long)total_tiles * (xcd + 1)) >> 3);
    } else { G = (int)gridDim.x; t = (int)blockIdx.x; t_end = total_tiles; }
    Coord cur = tile_coord(t), nxt = cur;
    load_w(0);
    load_x(cur, 0);
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * Co * 2));
    // tap tables of this workgroup's row blocks
    const int* __restrict__ gt = (const int*)p.up_taps;
    if constexpr (CK == 32) {
        if (tid < RB * NT) {
            const int code = gt[(rb0 + tid / NT) * NT + tid % NT];
            const int tap = code > 26 ? 13 : code;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            s_taps[tid] = ((dz * PLANE + dy * 18 + dx) * CKB) | (dx << 28);
        }
    } else {
        if (tid < RB * NT * 4) {                          // (rb, kg, g): lanes g = 0, 1 take tap 2 kg, g = 2, 3 tap 2 kg + 1
            const int rbk = tid >> 2, gg = tid & 3;
            const int code = gt[(rb0 + rbk / NT) * (2 * NT) + (rbk % NT) * 2 + (gg >> 1)];
            const int tap = code > 26 ? 13 : code;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            s_taps[tid] = (dz * PLANE + dy * 18 + dx) * CKB + (gg & 1) * 16;
        }
    }
    {   // the bias table of this workgroup's rows: every thread's loads requested before the first is stored (a load -> LDS store loop waits
        // for each in turn: seven L2 round trips in the prologue)
        constexpr int NBT = (27 * RB * 16 + 255) / 256;
        float bq[NBT];
#pragma unroll
        for (int u = 0; u < NBT; ++u) {
            const int i = tid + u * 256;
            const int cls = i / (RB * 16), lr = i - cls * (RB * 16);
            const int rbg = rb0 + lr / 16, r16 = lr & 15;
            const int co = Co >= 16 ? (rbg % nb) * 16 + r16 : (r16 & 7);
            bq[u] = (i < 27 * RB * 16 && p.up_btab != nullptr && rbg < p.rb_total) ? p.up_btab[cls * Co + co] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NBT; ++u)
            if (tid + u * 256 < 27 * RB * 16) s_bt[tid + u * 256] = bq[u];
    }
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(p.x_stats, (size_t)i, (size_t)st_n, st);
        float m, r;
        stats_to_mean_rstd_fast(st, p.inv_count_in, p.eps, m, r);
        s_scale[i] = r; s_shift[i] = -m * r;
    }
    int baddr[3];
    if constexpr (CK == 32) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) baddr[dx] = (wave * PLANE + col) * CKB + ((g ^ ((((col + dx) >> 2) & 1) << 1)) * 16);
    } else {
        baddr[0] = (wave * PLANE + col) * CKB;
    }
    const char* s_wl = s_w + lane * 16;
    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }
    bool first = true;
    if constexpr (W1) write_w();
    __syncthreads();
    // the per-lane LDS offset of every (row block, k-group)'s tap: tile independent, so it is read ONCE into registers — read inside the MFMA
    // loop, the table lookup put a second dependent LDS round trip in front of every k-group's B fragments
    int toff[RB * NT];
#pragma unroll
    for (int i = 0; i < RB * NT; ++i) {
        if constexpr (CK == 32) {
            const int code = s_taps[i];
            const int dx = code >> 28;
            toff[i] = (dx == 0 ? baddr[0] : (dx == 1 ? baddr[1] : baddr[2])) + (code & 0x0fffffff);
        } else {
            toff[i] = baddr[0] + s_taps[i * 4 + g];
        }
    }

    const int F_D = 2 * p.D, F_H = 2 * p.H, F_W = 2 * p.W;
    K4_TICK(0);
    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave;
        f32x4 acc[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < p.nch; ++ch) {
            if (!first) __syncthreads();
            K4_TICK(1);
            write_x(n, ch);
            if constexpr (!W1) write_w();
            first = false;
            K4_TICK(2);
            __syncthreads();
            K4_TICK(3);
            {
                const bool last_ch = ch + 1 == p.nch;
                const int tn = last_ch ? t + G : t;
                if (last_ch) nxt = tile_coord(tn);
                if (tn < t_end) {
                    if constexpr (!W1) load_w(last_ch ? 0 : ch + 1);
                    load_x(last_ch ? nxt : cur, last_ch ? 0 : ch + 1);
                }
            }
            auto read_kg = [&](int i, u32x4& a, u32x4 (&b)[4]) {     // i = rb * NT + kg
                a = *(const u32x4*)(s_wl + i * 1024);
                const int o = toff[i];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) b[cg] = *(const u32x4*)(s_tile + o + cg * 18 * CKB);
            };
            u32x4 fa[2], fb[2][4];
            K4_TICK(4);
            read_kg(0, fa[0], fb[0]);
#pragma unroll
            for (int i = 0; i < RB * NT; ++i) {
                if (i + 1 < RB * NT) read_kg(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc[i / NT][cg] = mfma16(fa[i & 1], fb[i & 1][cg], acc[i / NT][cg], (T*)nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
            K4_TICK(5);
        }

        // ---- epilogue: row (p, co) of coarse column v -> fine voxel 2 v + p ----
        // (address, boundary class and validity are built from per-tile / per-row-block / per-row pieces: the epilogue is VALU-bound,
        // ~45 instructions per stored fragment before the hoisting)
        const int ox = x0 + col;
        const bool zx_ok = oz < p.D && ox < p.W;
        const int e_tile = (((n * F_D + 2 * oz) * F_H + 2 * y0) * F_W + 2 * ox) * Co * 2;      // fine voxel (2 oz, 2 y0, 2 ox), channel 0
        const int e_row = 2 * F_W * Co * 2;                                                     // one coarse row = two fine rows
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int rbg = rb0 + rb;
            int pz, py, px, co0;
            if (Co >= 16) { const int pp = rbg / nb; pz = (pp >> 2) & 1; py = (pp >> 1) & 1; px = pp & 1; co0 = (rbg % nb) * 16 + 4 * g; }
            else { pz = (rbg >> 1) & 1; py = rbg & 1; px = g >> 1; co0 = (g & 1) * 4; }
            const bool rvalid = rbg < p.rb_total && zx_ok;
            const int fz = 2 * oz + pz, fx = 2 * ox + px;
            const int cz = fz == 0 ? 0 : (fz == F_D - 1 ? 2 : 1), cx = fx == 0 ? 0 : (fx == F_W - 1 ? 2 : 1);
            const float* bt0 = s_bt + (cz * 9 + cx) * (RB * 16) + rb * 16 + 4 * g;                  // + cy * 3 * RB * 16
            const int e_rb = e_tile + ((pz * F_H + py) * F_W + px) * Co * 2 + co0 * 2;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const int fy = 2 * (y0 + cg) + py;
                const int cy = fy == 0 ? 0 : (fy == F_H - 1 ? 2 : 1);
                const f32x4 bb = *(const f32x4*)(bt0 + cy * (3 * RB * 16));
                const bool valid = rvalid && y0 + cg < p.H;
                f32x2 lo, hi;
                lo[0] = acc[rb][cg][0] + bb[0]; lo[1] = acc[rb][cg][1] + bb[1];
                hi[0] = acc[rb][cg][2] + bb[2]; hi[1] = acc[rb][cg][3] + bb[3];
                i32x2 pk;
                pk[0] = (int)H16<T>::pack2(lo);
                pk[1] = (int)H16<T>::pack2(hi);
                vs_raw_buffer_store_b64(pk, yrsrc, valid ? e_rb + cg * e_row : -1, 0, 0);
                float v[4];
                v[0] = H16<T>::lo((unsigned int)pk[0]); v[1] = H16<T>::hi((unsigned int)pk[0]);
                v[2] = H16<T>::lo((unsigned int)pk[1]); v[3] = H16<T>::hi((unsigned int)pk[1]);
                if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[rb][r] += v[r]; ssq[rb][r] += v[r] * v[r]; }
            }
        }
        K4_TICK(6);
        if (p.y_stats != nullptr) {
            const bool flush = t + G >= t_end || nxt.n != n;
            if (flush) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = row16_sum(ssum[rb][r]), q = row16_sum(ssq[rb][r]);
                        if (col == 0) {
                            const int lr = rb * 16 + 4 * g + r;
                            s_red[(wave * 64 + lr) * 2 + 0] = s;
                            s_red[(wave * 64 + lr) * 2 + 1] = q;
                        }
                        ssum[rb][r] = 0.f; ssq[rb][r] = 0.f;
                    }
                __syncthreads();
                // rows of equal output channel (the parities) are folded first: one contribution per (workgroup, channel, statistic).
                // Two short passes (a single pass — 16 threads walking all 64 rows with the channel test — was 7 us per flush, two thirds of the
                // kernel): every (row, statistic) sums its four waves, then every (channel, statistic) its rows, which lie `ndist` apart.
                const int ndist = Co >= 16 ? (RB < nb ? RB : nb) * 16 : 8;
                float* s_row = s_red + 4 * 64 * 2;           // [RB * 16][2] (behind the per-wave partials)
                if (tid < RB * 16 * 2) {
                    const int lr = tid >> 1, st = tid & 1;
                    s_row[tid] = ((rb0 + lr / 16) < p.rb_total)
                                     ? (s_red[(0 * 64 + lr) * 2 + st] + s_red[(1 * 64 + lr) * 2 + st]) + (s_red[(2 * 64 + lr) * 2 + st] + s_red[(3 * 64 + lr) * 2 + st])
                                     : 0.f;
                }
                __syncthreads();
                if (tid < ndist * 2) {
                    const int ci = tid >> 1, st = tid & 1;
                    // local rows of this channel: Co = 8: ci, ci + 8, ...; Co >= 16: row (ci & 15) of the row blocks whose co-block is ci / 16 (every nb-th, or one)
                    double tot = 0.0;
                    int co;
                    if (Co >= 16) {
                        const int cbl = ci >> 4;                                          // index among this workgroup's distinct co-blocks
                        co = (((rb0 % nb) + cbl) % nb) * 16 + (ci & 15);
                        for (int rb = cbl; rb < RB; rb += (RB < nb ? RB : nb)) tot += (double)s_row[(rb * 16 + (ci & 15)) * 2 + st];
                    } else {
                        co = ci;
                        for (int lr = ci; lr < RB * 16; lr += 8) tot += (double)s_row[lr * 2 + st];
                    }
                    stat_add(p.y_stats, (size_t)n * Co + co, (size_t)p.N * Co, st, tot);
                }
                if (t + G < t_end) __syncthreads();
            }
        }
        cur = nxt;
        K4_TICK(1);            // (diagnostic build: the statistics flush is booked under slot 1)
    }
    K4_TICK_FLUSH;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// backward-data: coarse gradient of the block's input from the fine gradient of the 3x3x3 conv's raw output
// ---------------------------------------------------------------------------------------------------------------------------------------
// NT: taps per 32-"channel" chunk of the space-to-depth view (8: one parity per chunk, Co >= 32; 12: Co = 16, a chunk holds both x parities;
// 18: Co = 8, a chunk holds the four (py, px) parities).  The view's channel (p, co) of coarse voxel v' is channel co of fine voxel 2v' + p.
template <int MT, int NT, bool SUMS, typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k4g_kernel(const G1Params p) {
    constexpr int TV = K4Geom::TV, PLANE = K4Geom::PLANE;
    constexpr int RB = MT / 16, CKB = 64, U = 4, NU = TV * U, NIT = (NU + 255) / 256;
    constexpr int NWF = RB * NT * 64, NWI = (NWF + 255) / 256;
    constexpr int TILE_BYTES = ((NU + 255) / 256) * 256 * 16, W_BYTES = ((NWF + 255) / 256) * 256 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K4_LDS_RED);
    int* s_taps = (int*)(smem + K4_LDS_TAPS);            // [chunk][NT] (chunks x NT <= 512)
    char* s_tile = smem + K4_LDS_TILE;
    char* s_w = s_tile + TILE_BYTES;
    float* s_mkm = (float*)(s_w + W_BYTES);
    float* s_mkr = s_mkm + p.N * p.M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    const int Co = p.up_co;
    const int FH = 2 * p.H, FW = 2 * p.W;                // p.D/H/W: the COARSE grid (output); the input gradient lives on 2D x 2H x 2W x Co
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * Co * 2));
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    const int st_n = SUMS ? p.N * p.M : 0;
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(p.mask_stats, (size_t)tid, (size_t)st_n, st_pre);

    // fragment b of this thread: 16-byte part `part` of (halo voxel tv_b, chunk): part -> (parity bits inside the chunk, 8-channel group)
    const int part = tid & 3;
    int part_off;                                         // bytes
    if (Co >= 16) part_off = part * 16;                                               // Co >= 32: 4 x 8 channels; Co = 16: (px, half) — contiguous either way
    else part_off = (part & 1) * 16 + (part >> 1) * (FW * Co * 2);                     // Co = 8: part = (py, px)
    int rel_off[NIT], tzyx[NIT];
    const int lds_w0 = (tid >> 2) * CKB;
    unsigned int swzbits = 0;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u >> 2;
        const int tx_ = tv % 18, ty_ = (tv / 18) % 6, tz_ = tv / PLANE;
        rel_off[b] = (((2 * tz_) * FH + 2 * ty_) * FW + 2 * tx_) * Co * 2 + part_off;
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;
        swzbits |= (unsigned int)((tx_ >> 2) & 1) << b;
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int f = tid + i * 256;
        if (f > NWF - 1) f = NWF - 1;
        const int rb = f / (NT * 64), r = f - rb * (NT * 64);
        w_off[i] = (rb0 + rb) * (p.nch * NT * 64) + r;
    }
    u32x4 xv[NIT], wv[NWI];
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {
        Coord c;
        c.n = fdiv(t, p.fd_m[0], p.fd_s[0]);
        const int tl = t - c.n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], p.fd_s[1]);
        const int r = tl - tz * (p.txn * p.tyn);
        const int ty = fdiv(r, p.fd_m[2], p.fd_s[2]);
        c.z0 = tz * 4; c.y0 = ty * 4; c.x0 = (r - ty * p.txn) * 16;
        return c;
    };
    auto chunk_off = [&](int ch) {                        // byte offset of the chunk's first sub-voxel / channel group inside a coarse voxel's 2x2x2 block
        if (Co >= 32) {
            const int cb = Co / 32, pp = ch / cb, c32 = ch - pp * cb;
            return ((((pp >> 2) & 1) * FH + ((pp >> 1) & 1)) * FW + (pp & 1)) * Co * 2 + c32 * 64;
        }
        if (Co == 16) return ((((ch >> 1) & 1) * FH + (ch & 1)) * FW) * Co * 2;          // chunk = (pz, py)
        return (ch * FH * FW) * Co * 2;                                                     // Co = 8: chunk = pz
    };
    auto load_w = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + ch * (NT * 64)];
    };
    auto load_x = [&](const Coord& c, int ch) {
        const int base = (((c.n * 2 * p.D + 2 * (c.z0 - 1)) * FH + 2 * (c.y0 - 1)) * FW + 2 * (c.x0 - 1)) * Co * 2 + chunk_off(ch);
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int gz = c.z0 - 1 + (tzyx[b] & 0xff), gy = c.y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = c.x0 - 1 + (tzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
        }
    };
    auto write_x = [&]() {
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int pw = part ^ (int)(((swzbits >> b) & 1u) << 1);
            *(u32x4*)(s_tile + lds_w0 + b * 4096 + pw * 16) = xv[b];
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i) *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    };

    int t, t_end, G;
    if (((int)gridDim.x & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7;
        G = (int)gridDim.x >> 3;
        t = (int)(((long long)total_tiles * xcd) >> 3) + ((int)blockIdx.x >> 3);
        t_end = (int)(((long long)total_tiles * (xcd + 1)) >> 3);
    } else { G = (int)gridDim.x; t = (int)blockIdx.x; t_end = total_tiles; }
    Coord cur = tile_coord(t), nxt = cur;
    load_w(0);
    load_x(cur, 0);
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 2));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 2));
    const int* __restrict__ gt = (const int*)p.up_taps;
    for (int i = tid; i < p.nch * NT; i += 256) {
        const int code = gt[i];
        const int tap = code > 26 ? 13 : code;
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        s_taps[i] = ((dz * PLANE + dy * 18 + dx) * CKB) | (dx << 28);
    }
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(p.mask_stats, (size_t)i, (size_t)st_n, st);
        float m, r;
        stats_to_mean_rstd_fast(st, p.inv_count_out, p.eps, m, r);
        s_mkm[i] = m; s_mkr[i] = r;
    }
    int baddr[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) baddr[dx] = (wave * PLANE + col) * CKB + ((g ^ ((((col + dx) >> 2) & 1) << 1)) * 16);
    const char* s_wl = s_w + lane * 16;
    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }
    bool first = true;
    __syncthreads();

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave;
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + x0 + col) * p.M + rb0 * 16 + 4 * g) * 2;
        const bool zx_ok = oz < p.D && x0 + col < p.W;
        f32x4 acc[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 mk[RB][4];

        for (int ch = 0; ch < p.nch; ++ch) {
            // this chunk's tap offsets: requested before the stage is written, in registers by the time the MFMA loop starts
            int toff[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int code = s_taps[ch * NT + j];
                const int dx = code >> 28;
                toff[j] = (dx == 0 ? baddr[0] : (dx == 1 ? baddr[1] : baddr[2])) + (code & 0x0fffffff);
            }
            if (!first) __syncthreads();
            write_x();
            write_w();
            first = false;
            __syncthreads();
            const bool last_ch = ch + 1 == p.nch;
            if (SUMS && last_ch) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) {
                        const bool valid = zx_ok && y0 + cg < p.H && (rb0 + rb) * 16 + 4 * g < p.M;
                        mk[rb][cg] = __builtin_bit_cast(u32x2, vs_raw_buffer_load_b64(mrsrc, valid ? ebase + cg * p.W * p.M * 2 + rb * 32 : -1, 0, 0));
                    }
            }
            {
                const int tn = last_ch ? t + G : t;
                if (last_ch) nxt = tile_coord(tn);
                if (tn < t_end) {
                    load_w(last_ch ? 0 : ch + 1);
                    load_x(last_ch ? nxt : cur, last_ch ? 0 : ch + 1);
                }
            }
            auto read_kg = [&](int j, u32x4 (&a)[RB], u32x4 (&b)[4]) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = *(const u32x4*)(s_wl + (rb * NT + j) * 1024);
                const int o = toff[j];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) b[cg] = *(const u32x4*)(s_tile + o + cg * 18 * CKB);
            };
            u32x4 fa[2][RB], fb[2][4];
            read_kg(0, fa[0], fb[0]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j + 1 < NT) read_kg(j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = mfma16(fa[j & 1][rb], fb[j & 1][cg], acc[rb][cg], (T*)nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const bool rvalid = (rb0 + rb) * 16 + 4 * g < p.M;
            float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f};
            if (SUMS && rvalid) {
                const int row = (rb0 + rb) * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * p.M + row + r]; mr[r] = s_mkr[n * p.M + row + r]; }
            }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const bool valid = rvalid && zx_ok && y0 + cg < p.H;
                f32x2 lo, hi;
                lo[0] = acc[rb][cg][0]; lo[1] = acc[rb][cg][1];
                hi[0] = acc[rb][cg][2]; hi[1] = acc[rb][cg][3];
                i32x2 pk;
                pk[0] = (int)H16<T>::pack2(lo);
                pk[1] = (int)H16<T>::pack2(hi);
                vs_raw_buffer_store_b64(pk, yrsrc, valid ? ebase + cg * p.W * p.M * 2 + rb * 32 : -1, 0, 0);
                if (SUMS) {
                    float v[4];
                    v[0] = H16<T>::lo((unsigned int)pk[0]); v[1] = H16<T>::hi((unsigned int)pk[0]);
                    v[2] = H16<T>::lo((unsigned int)pk[1]); v[3] = H16<T>::hi((unsigned int)pk[1]);
                    if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                    const u32x2 xx = mk[rb][cg];
                    float xv4[4];
                    xv4[0] = H16<T>::lo(xx[0]); xv4[1] = H16<T>::hi(xx[0]);
                    xv4[2] = H16<T>::lo(xx[1]); xv4[3] = H16<T>::hi(xx[1]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (xv4[r] - mm[r]) * mr[r];
                        const float gm = xh > 0.f ? v[r] : 0.f;
                        ssum[rb][r] += gm; ssq[rb][r] += gm * xh;
                    }
                }
            }
        }
        if (SUMS) {
            const bool flush = t + G >= t_end || nxt.n != n;
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
                    if (row < p.M) {
                        const double tot = (double)s_red[(0 * 64 + lr) * 2 + st] + (double)s_red[(1 * 64 + lr) * 2 + st] +
                                           (double)s_red[(2 * 64 + lr) * 2 + st] + (double)s_red[(3 * 64 + lr) * 2 + st];
                        stat_add(p.sums, (size_t)n * p.M + row, (size_t)p.N * p.M, st, tot);
                    }
                }
                if (t + G < t_end) __syncthreads();
            }
        }
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------------------------
static inline void k4_fastdiv(int d, unsigned int& m, unsigned int& s) {
    s = 0;
    while ((1ll << s) < d) ++s;
    m = (unsigned int)((((1ull << (32 + s)) + (unsigned long long)d - 1) / (unsigned long long)d) - (1ull << 32));
}

static inline int k4_grid_x(int tiles_total, int row_tiles) {
    const int per_cu = vs_cfg().up_wgs_per_cu;      // tuning knob
    int wg = 256 * per_cu / (row_tiles < per_cu ? row_tiles : per_cu);
    if (wg < 256) wg = 256;
    return tiles_total < wg ? tiles_total : wg;
}

template <typename T, int CK, int RB, bool HS, bool W1 = (CK == 16)>
static int k4t_launch_t(G1Params p, hipStream_t stream) {
    constexpr int NT = CK == 32 ? 8 : 6, NU = 648 * (CK * 2 / 16), NWF = RB * NT * 64;
    const size_t lds = K4_LDS_TILE + (size_t)((NU + 255) / 256) * 4096 + (size_t)((NWF + 255) / 256) * 4096 + (size_t)2 * p.N * p.C * 4 + (size_t)27 * RB * 16 * 4;
    if (lds > 160 * 1024) return VS_ESHAPE;
    const int row_tiles = (p.rb_total + RB - 1) / RB;
    k4_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k4_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k4_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    auto kern = k4t_kernel<CK, RB, HS, T, W1>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(k4_grid_x(p.tiles_per_sample * p.N, row_tiles), row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T>
static int k4t_launch(const G1Params& p, int rb, hipStream_t s) {
    const bool hs = p.x_stats != nullptr;
    if (p.C == 16) {
        if (p.up_co != 8) return VS_ESHAPE;
        if (rb == 2) return hs ? k4t_launch_t<T, 16, 2, true>(p, s) : k4t_launch_t<T, 16, 2, false>(p, s);
        return hs ? k4t_launch_t<T, 16, 4, true>(p, s) : k4t_launch_t<T, 16, 4, false>(p, s);
    }
    if (p.C % 32 || p.up_co % 16) return VS_ESHAPE;
    if (p.nch == 1) {
        if (rb == 4) return hs ? k4t_launch_t<T, 32, 4, true, true>(p, s) : k4t_launch_t<T, 32, 4, false, true>(p, s);
        return hs ? k4t_launch_t<T, 32, 2, true, true>(p, s) : k4t_launch_t<T, 32, 2, false, true>(p, s);
    }
    if (rb == 4) return hs ? k4t_launch_t<T, 32, 4, true>(p, s) : k4t_launch_t<T, 32, 4, false>(p, s);
    return hs ? k4t_launch_t<T, 32, 2, true>(p, s) : k4t_launch_t<T, 32, 2, false>(p, s);
}

template <typename T, int MT, int NT, bool SUMS>
static int k4g_launch_t(G1Params p, hipStream_t stream) {
    constexpr int RB = MT / 16, NWF = RB * NT * 64;
    const size_t lds = K4_LDS_TILE + (size_t)((648 * 4 + 255) / 256) * 4096 + (size_t)((NWF + 255) / 256) * 4096 + (size_t)2 * p.N * p.M * 4;
    if (lds > 160 * 1024 || p.nch * NT > 512) return VS_ESHAPE;
    const int row_tiles = (p.rb_total + RB - 1) / RB;
    k4_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k4_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k4_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    auto kern = k4g_kernel<MT, NT, SUMS, T>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(k4_grid_x(p.tiles_per_sample * p.N, row_tiles), row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T>
static int k4g_launch(const G1Params& p, int mt, hipStream_t s) {
    const int nt = p.up_co >= 32 ? 8 : (p.up_co == 16 ? 12 : 18);
    const bool sums = p.sums != nullptr;
#define K4G_CASE(MTV, NTV)                                                                             \
    if (mt == MTV && nt == NTV) return sums ? k4g_launch_t<T, MTV, NTV, true>(p, s) : k4g_launch_t<T, MTV, NTV, false>(p, s);
    K4G_CASE(16, 8) K4G_CASE(32, 8) K4G_CASE(16, 12) K4G_CASE(32, 12) K4G_CASE(16, 18)
#undef K4G_CASE
    return VS_ESHAPE;
}
