// 3x3x3 (pad 1) convolution in the fp32 PARITY mode on the bf16 matrix cores: forward and backward-data.
//
// The exact-f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s dense), which made every 3x3x3 launch of
// the parity mode MFMA-cycle bound (k3_kernel, igemm_k3.h: 100-135 us for an 8 -> 8 layer at 96^3 against ~25 us of HBM time).  Here each fp32
// operand is split into three bf16 limbs,  x = x0 + x1 + x2  (x0 = x rounded to bf16, x1 = x - x0 rounded to bf16, x2 = the rest: 8 + 8 + 8
// significant bits, every subtraction exact, the sum exact to 2^-25 |x|), and the product is formed from the six limb products of weight >= 2^-16,
//     x*w  ~=  x0*w0 + x0*w1 + x1*w0 + x0*w2 + x1*w1 + x2*w0                      (dropped: x1*w2 + x2*w1 + x2*w2 <= 3 * 2^-24 |x*w|)
// each of which is EXACT in the fp32 accumulator of v_mfma_f32_16x16x32_bf16 (8 x 8 significant bits).  Six bf16 MFMAs replace one exact-f32
// group at 16x the rate: 2.7x fewer matrix cycles, and the error of the dropped terms (2e-7 relative) is that of one fp32 rounding — the
// results stay inside every fp32-mode tolerance of the test-suite (2e-5) and far inside north_star's 1e-3.
//
// Structure: k3b_kernel's (igemm_k3b.h) — persistent XCD-aware tile walk, a stage = (4x4x16 tile, 8- or 16-channel chunk) fetched whole into
// registers one stage ahead through bounds-checked buffer loads, normalise+ReLU of a lazy input in fp32 while staging — except that the staged
// halo tile is written to LDS as three bf16 limb PLANES and the workgroup's weight block holds three limb fragments per k-group (pack.hip:
// VS_F32X3 image).  Storage stays fp32 on both sides: 16-byte fragments are 4 channels of one voxel.
#pragma once
#include <stdlib.h>
#include "igemm.h"

#include "chain.h"
#define K3X_LDS_RED 0          // float[4][64][2]
#define K3X_LDS_TAPS 2048      // int[64]: byte offset of (k-group, lane group)'s tap in a limb plane
#define K3X_LDS_TILE 2304      // three limb planes, the weight block, then the per-(n,c) tables

template <int CK, int MT, int YT = 4>
struct K3XGeom {
    static constexpr int RB = MT / 16;
    static constexpr int TV = 6 * (YT + 2) * 18;                             // staged halo voxels (4 x YT x 16 tile)
    static constexpr int U = CK / 4;                                         // fp32 fragments (4 channels) per staged voxel
    static constexpr int NIT = (TV * U + 255) / 256;                         // fragments per thread per stage
    static constexpr int CKB2 = CK * 2;                                      // bytes per voxel in one limb plane
    static constexpr int PLANE_BYTES = NIT * 256 * 8;                        // padded: every thread stores all its fragments (8 bytes per plane each)
    static constexpr int NKGC = (27 * CK + 31) / 32;                         // 32-wide k-groups per chunk: 7 (CK = 8: four taps each) / 14 (CK = 16: two taps each)
    static constexpr int NWF = RB * NKGC * 3 * 64;                           // 16-byte weight fragments per chunk per workgroup (three limbs)
    static constexpr int NWI = (NWF + 255) / 256;
    static constexpr int W_BYTES = NWI * 256 * 16;
};

// SUMS: backward-data use (fused IN-backward sums of the output against the mask tensor); HS: the input is a lazy activation; MULTI: more than one
// channel chunk (the weight block is re-staged per chunk) — all compile-time, like every condition on the staging path (igemm_k3b.h)
// FA (backward-data use; round 5, the 16-channel layers of the 48^3 level): the input gradient arrives UN-applied, as for k3xt_kernel's fused apply below — p.x = g = dL/da
// of the lazy activation a = relu(norm(p.fa_x)), statistics p.x_stats, IN-backward sums p.fa_sums; the apply runs in fp32 on the staged fragments before the
// limb split, centre voxels also go to p.fa_dx when given (one row-block workgroup per tile stores them)
// YT (round 6): tile extent in y — 4, or 2 / 1 for the under-filled launches of the 12^3-class levels (igemm_k3_h16.inc says why; dispatch in igemm_k3x.hip)
// EA (round 6): the epilogue apply of igemm_k3b.h for the parity mode — backward-data with fused sums whose workgroups (one tile each, all resident) wait for their
// SAMPLE's sums and store the applied gradient themselves; fp32 values throughout, so the result equals the standalone apply's bit for bit (deterministic build)
template <int CK, int MT, int EPI, bool SUMS, bool HS, bool MULTI, bool FA = false, int YT = 4, bool EA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CK == 8 ? 2 : 1))) void k3x_kernel(const G1Params p) {
    using GEO = K3XGeom<CK, MT, YT>;
    static_assert(!EA || (SUMS && !FA && !HS && EPI == EPI_RAW), "epilogue apply: backward-data kernels with fused sums");
    static_assert(YT == 4 || ((YT == 2 || YT == 1) && !FA && EPI == EPI_RAW), "short tiles: plain / fused-sums launches");
    static_assert(CK == 8 || CK == 16, "chunk width");
    static_assert(!FA || (!HS && EPI == EPI_RAW), "fused apply: backward-data use");
    constexpr int TV = GEO::TV, PLANE = (YT + 2) * 18, U = GEO::U, NIT = GEO::NIT, CKB2 = GEO::CKB2, RB = GEO::RB, NKGC = GEO::NKGC;
    constexpr int NU = TV * U, NWF = GEO::NWF, NWI = GEO::NWI, PB = GEO::PLANE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3X_LDS_RED);
    int* s_taps = (int*)(smem + K3X_LDS_TAPS);
    char* s_tile = smem + K3X_LDS_TILE;
    char* s_w = s_tile + 3 * PB;
    float* s_mean = (float*)(s_w + GEO::W_BYTES);        // mean / rstd of the lazy input, [N*C] each
    float* s_rstd = s_mean + p.N * p.C;
    float* s_mkm = s_rstd + p.N * p.C;                   // mean / rstd of the mask tensor's channels (fused IN-bwd sums)
    float* s_mkr = s_mkm + p.N * p.M;
    float* s_fa = s_rstd + p.N * p.C + (SUMS ? 2 * p.N * p.M : 0);   // FA: rstd, -mean*rstd, m1, m2 of the input gradient's activation, [N*C] each

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 4));
    const i32x4 frsrc = make_rsrc(FA ? p.fa_x : p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 4));
    const i32x4 dxrsrc = make_rsrc(FA && p.fa_dx != nullptr ? p.fa_dx : p.x, (FA && p.fa_dx != nullptr) ? (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 4) : 0u);
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    // the (sum, sumsq) pair this thread turns into a table entry is requested first of all (oldest load in the queue)
    const double* st_src = SUMS ? p.mask_stats : p.x_stats;
    const int st_n = SUMS ? p.N * p.M : (HS ? p.N * p.C : 0);
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(st_src, (size_t)tid, (size_t)st_n, st_pre);
    double fa_pre[2][2] = {{0.0, 1.0}, {0.0, 0.0}};      // FA: waves 1.. request the activation's (sum, sumsq) and (sum g*mask, sum g*mask*xhat) pairs
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * p.C) {
            stat_load(p.x_stats, (size_t)(tid - 64), (size_t)p.N * p.C, fa_pre[0]);
            stat_load(p.fa_sums, (size_t)(tid - 64), (size_t)p.N * p.C, fa_pre[1]);
        }
    }

    // ---- per-thread stage geometry (tile independent): fragment b = channels 4*part .. 4*part+3 of tile voxel tv_b ----
    const int part = tid % U;
    int rel_off[NIT], tzyx[NIT];
    unsigned int cbits = 0;                              // FA: fragment b belongs to a centre (non-halo) voxel of the tile
    const int lds_w0 = (tid / U) * CKB2 + part * 8;       // + b * 2048 (tv advances by 256 / U voxels of CKB2 bytes per b), + limb * PB
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u / U;
        const int tx_ = tv % 18, ty_ = (tv / 18) % (YT + 2), tz_ = tv / PLANE;
        rel_off[b] = (((tz_ * p.H + ty_) * p.W + tx_) * p.C + part * 4) * 4;              // bytes from the tile's (0,0,0) halo voxel
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;                 // out-of-list fragments fail every bounds test
        cbits |= (u < NU && tz_ >= 1 && tz_ <= 4 && ty_ >= 1 && ty_ <= YT && tx_ >= 1 && tx_ <= 16) ? (1u << b) : 0u;
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int f = tid + i * 256;
        if (f > NWF - 1) f = NWF - 1;
        const int rb = f / (NKGC * 192), r = f - rb * (NKGC * 192);
        w_off[i] = (rb0 + rb) * (p.nch * NKGC * 192) + r;                                  // + ch * NKGC * 192
    }

    u32x4 xv[NIT], wv[NWI], fv[FA ? NIT : 1];
    unsigned int okbits = 0;
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {
        Coord c;
        c.n = fdiv(t, p.fd_m[0], p.fd_s[0]);
        const int tl = t - c.n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], p.fd_s[1]);
        const int r = tl - tz * (p.txn * p.tyn);
        const int ty = fdiv(r, p.fd_m[2], p.fd_s[2]);
        c.z0 = tz * 4; c.y0 = ty * YT; c.x0 = (r - ty * p.txn) * 16;
        return c;
    };
    auto load_w = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + ch * (NKGC * 192)];
    };
    auto load_x = [&](const Coord& c, int ch) {
        const int base = ((((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * p.C + ch * CK) * 4;
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
    auto write_x_fa = [&](const Coord& c, int ch) {     // FA: apply pass in fp32 on the staged fragments, then the limb split [+ the applied gradient of the centre voxels]
        float rr[4], ss[4], aa[4], bb[4];
        const int c0 = c.n * p.C + ch * CK + part * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rr[j] = s_fa[0 * p.N * p.C + c0 + j];
            ss[j] = s_fa[1 * p.N * p.C + c0 + j];
            aa[j] = s_fa[2 * p.N * p.C + c0 + j];
            bb[j] = s_fa[3 * p.N * p.C + c0 + j];
        }
        const int base = ((((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * p.C + ch * CK) * 4;
        const bool store_dx = p.fa_dx != nullptr && blockIdx.y == 0;          // workgroup-uniform: once per tile
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const bool ok = (okbits >> b) & 1u;           // out-of-volume halo voxels: the gradient is zero-padded
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gq = __uint_as_float(xv[b][j]);
                const float xh = __uint_as_float(fv[b][j]) * rr[j] + ss[j];
                const float gm = xh > 0.f ? gq : 0.f;
                const float d = rr[j] * (gm - aa[j] - xh * bb[j]);
                v[j] = ok ? d : 0.f;
            }
            unsigned int lm[3][2];
            vs_limb_split4(v, lm);
#pragma unroll
            for (int l = 0; l < 3; ++l) *(u32x2*)(s_tile + l * PB + lds_w0 + b * 2048) = u32x2{lm[l][0], lm[l][1]};
            if (store_dx)
                vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), dxrsrc, (ok && ((cbits >> b) & 1u)) ? base + rel_off[b] : -1, 0, 0);
        }
    };
    auto write_x = [&](int n, int ch) {                   // normalise + ReLU (fp32), split into limbs, three 8-byte stores
        float mn[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (HS) {
            const int c0 = n * p.C + ch * CK + part * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { mn[j] = s_mean[c0 + j]; rs[j] = s_rstd[c0 + j]; }
        }
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(xv[b][j]);
            if constexpr (HS) {
                const bool ok = (okbits >> b) & 1u;       // zero padding applies to the normalised activation
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = (v[j] - mn[j]) * rs[j];
                    v[j] = ok ? fmaxf(t, 0.f) : 0.f;
                }
            }
            unsigned int lm[3][2];
            vs_limb_split4(v, lm);
#pragma unroll
            for (int l = 0; l < 3; ++l) *(u32x2*)(s_tile + l * PB + lds_w0 + b * 2048) = u32x2{lm[l][0], lm[l][1]};
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i) *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    };

    // ---- XCD-aware persistent walk (igemm_k3b.h) ----
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
    float bv[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (rb0 + rb) * 16 + 4 * g + r;
            bv[rb][r] = (p.bias != nullptr && row < (EPI == EPI_SOFTMAX2 ? 2 : p.M)) ? p.bias[row] : 0.f;
        }
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 4));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 4));
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(st_src, (size_t)i, (size_t)st_n, st);
        float m, r;
        pair_to_mean_rstd(st, SUMS ? p.inv_count_out : p.inv_count_in, p.eps, m, r);     // the fp32 mode's tables: fp64 sqrt, as in k3_kernel / g1_kernel
        if constexpr (SUMS) { s_mkm[i] = m; s_mkr[i] = r; }
        else { s_mean[i] = m; s_rstd[i] = r; }
    }
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * p.C) {
            const int i = tid - 64;
            float m, r;
            pair_to_mean_rstd(fa_pre[0], p.inv_count_in, p.eps, m, r);
            s_fa[0 * p.N * p.C + i] = r;
            s_fa[1 * p.N * p.C + i] = -m * r;
            s_fa[2 * p.N * p.C + i] = (float)(fa_pre[1][0] * p.inv_count_in);
            s_fa[3 * p.N * p.C + i] = (float)(fa_pre[1][1] * p.inv_count_in);
        }
    }
    // B fragment of (k-group kg, lane group g, column voxel (wave, cg, col)): limb plane + (wave * PLANE + cg * 18 + col) * CKB2 + s_taps[4 kg + g]
    if (tid < NKGC * 4) {
        int tap = (tid >> 2) * (32 / CK) + ((tid & 3) * 8) / CK;
        if (tap > 26) tap = 13;                          // padded taps read the centre voxel (their weights are zero)
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        s_taps[tid] = (dz * PLANE + dy * 18 + dx) * CKB2 + (((tid & 3) * 8) % CK) * 2;
    }
    const int baddr = (wave * PLANE + col) * CKB2;
    const char* s_wl = s_w + lane * 16;

    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[rb][r]));      // the bias' wait belongs to the prologue (igemm_k3b.h)
    bool first = true;
    if constexpr (!MULTI) write_w();
    __syncthreads();                                     // tables visible

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave;
        // byte offset of output voxel (n, oz, y0 + cg, x0 + col), row 4g of row block rb: ebase + cg * W*M*4 + rb * 64; -1 = dropped
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + x0 + col) * p.M + rb0 * 16 + 4 * g) * 4;
        const bool zx_ok = oz < p.D && x0 + col < p.W;
        // two accumulators per output tile: the leading products x0*w0 in `acc`, the five products of weight <= 2^-8 in `acl`, added once in the
        // epilogue.  In one accumulator the big running sum was rounded by all six MFMAs of every k-group; measured on the layer tests (max error of y
        // against CPU fp32 autograd): 1.2e-6 with one accumulator, the exact-f32 kernels' 4e-7 with two
        f32x4 acc[RB][YT], acl[RB][YT];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) { acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        u32x4 mk[RB][YT];                                 // mask tensor values under this tile's outputs (fused IN-bwd sums)
        f32x4 vkeep[EA ? RB : 1][EA ? YT : 1];            // EA: this tile's outputs, kept for the apply

        for (int ch = 0; ch < p.nch; ++ch) {
            if (!first) __syncthreads();                 // every wave is done reading the previous stage
            if constexpr (FA) write_x_fa(cur, ch); else write_x(n, ch);
            if constexpr (MULTI) write_w();
            first = false;
            __syncthreads();
            const bool last_ch = ch + 1 == p.nch;
            if constexpr (EPI == EPI_RAW && SUMS) {
                if (last_ch) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cg = 0; cg < YT; ++cg) {
                            const bool valid = zx_ok && y0 + cg < p.H && (rb0 + rb) * 16 + 4 * g < p.M;
                            mk[rb][cg] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(mrsrc, valid ? ebase + cg * p.W * p.M * 4 + rb * 64 : -1, 0, 0));
                        }
                }
            }
            {
                const int tn = last_ch ? t + G : t;
                if (last_ch) nxt = tile_coord(tn);
                if (tn < t_end) {
                    if constexpr (MULTI) load_w(last_ch ? 0 : ch + 1);
                    load_x(last_ch ? nxt : cur, last_ch ? 0 : ch + 1);
                }
            }
            // ---- multiply this stage out of LDS: per k-group three A limbs per row block, three B limbs per column group, six MFMAs per pair ----
            auto read_kg = [&](int kg, u32x4 (&a)[RB][3], u32x4 (&b)[YT][3]) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int j = 0; j < 3; ++j) a[rb][j] = *(const u32x4*)(s_wl + ((rb * NKGC + kg) * 3 + j) * 1024);
                const int o = baddr + s_taps[kg * 4 + g];
#pragma unroll
                for (int cg = 0; cg < YT; ++cg)
#pragma unroll
                    for (int i = 0; i < 3; ++i) b[cg][i] = *(const u32x4*)(s_tile + i * PB + o + cg * 18 * CKB2);
            };
            u32x4 fa[2][RB][3], fb[2][YT][3];
            read_kg(0, fa[0], fb[0]);
#pragma unroll
            for (int kg = 0; kg < NKGC; ++kg) {
                if (kg + 1 < NKGC) read_kg(kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
                // limb pairs (activation limb i, weight limb j), smallest products first; four independent accumulators between dependent MFMAs
                constexpr int PI[6] = {2, 1, 0, 1, 0, 0}, PJ[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 5; ++q)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cg = 0; cg < YT; ++cg)
                            acl[rb][cg] = mfma16(fa[kg & 1][rb][PJ[q]], fb[kg & 1][cg][PI[q]], acl[rb][cg], (unsigned short*)nullptr);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < YT; ++cg)
                        acc[rb][cg] = mfma16(fa[kg & 1][rb][0], fb[kg & 1][cg][0], acc[rb][cg], (unsigned short*)nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of this tile ----
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) acc[rb][cg] += acl[rb][cg];
        if constexpr (EPI == EPI_SOFTMAX2) {
            if (g == 0) {
                const float b0 = bv[0][0], b1 = bv[0][1];
                const size_t V = (size_t)p.D * p.H * p.W;
#pragma unroll
                for (int cg = 0; cg < YT; ++cg) {
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
                for (int cg = 0; cg < YT; ++cg) {
                    const bool valid = rvalid && zx_ok && y0 + cg < p.H;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[rb][cg][r] + bv[rb][r];
                    if constexpr (EA) vkeep[rb][cg] = f32x4{v[0], v[1], v[2], v[3]};
                    else vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), yrsrc, valid ? ebase + cg * p.W * p.M * 4 + rb * 64 : -1, 0, 0);
                    if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                    if constexpr (SUMS) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (__uint_as_float(mk[rb][cg][r]) - mm[r]) * mr[r];
                            const float gm = xh > 0.f ? v[r] : 0.f;
                            ssum[rb][r] += gm; ssq[rb][r] += gm * xh;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[rb][r] += v[r]; ssq[rb][r] += v[r] * v[r]; }
                    }
                }
            }
            double* const red_dst = SUMS ? p.sums : p.y_stats;
            if (red_dst != nullptr) {
                const bool flush = t + G >= t_end || nxt.n != n;       // workgroup-uniform
                if (flush) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s = ssum[rb][r], q = ssq[rb][r];
                            s = row16_sum(s); q = row16_sum(q);
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
                            stat_add(red_dst, (size_t)n * p.M + row, (size_t)p.N * p.M, st, tot);
                        }
                    }
                    if (t + G < t_end) __syncthreads();     // s_red is reused by a later flush
                }
            }
            if constexpr (EA) {
                unsigned int* ctr = p.ea_sync + (size_t)n * 256;          // 8 shards of 128 bytes per sample (chain.h)
                chain_arrive8(ctr);
                chain_wait8(ctr, (unsigned int)p.ea_items, p.ea_fault);
                if (tid < MT) {
                    const int row = rb0 * 16 + tid;
                    float m = 0.f, r = 1.f, a = 0.f, b = 0.f;
                    if (row < p.M) {
                        stats_to_mean_rstd(p.mask_stats, (size_t)n * p.M + row, (size_t)p.N * p.M, p.inv_count_out, p.eps, m, r);     // the standalone apply's exact form
                        double sv[2];
                        stat_load_sc1(p.sums, (size_t)n * p.M + row, (size_t)p.N * p.M, sv);
                        a = (float)(sv[0] * p.inv_count_out);
                        b = (float)(sv[1] * p.inv_count_out);
                    }
                    *(f32x4*)(s_red + tid * 4) = f32x4{m, r, a, b};
                }
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const bool rvalid = (rb0 + rb) * 16 + 4 * g < p.M;
                    f32x4 tb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb[r] = *(const f32x4*)(s_red + (rb * 16 + 4 * g + r) * 4);
#pragma unroll
                    for (int cg = 0; cg < YT; ++cg) {
                        const bool valid = rvalid && zx_ok && y0 + cg < p.H;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = vs_in_bwd_apply1(vkeep[rb][cg][r], __uint_as_float(mk[rb][cg][r]), tb[r][0], tb[r][1], tb[r][2], tb[r][3]);
                        }
                        vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{o[0], o[1], o[2], o[3]}), yrsrc, valid ? ebase + cg * p.W * p.M * 4 + rb * 64 : -1, 0, 0);
                    }
                }
            }
        }
        cur = nxt;
    }
}

// workgroups of one k3x_kernel<8, 16, .., EA> launch that are certainly resident together: ONE per CU — the 4 x 4 x 16 instantiation holds 226 + 48 registers
// (one wave per SIMD), whatever the LDS would allow (the first version counted LDS only: 288 workgroups waited for 32 that could not start, and the fault word said so)
static inline int k3x_ea_max_wgs(int, int, int) { return 256; }

// (m, s) with n / d == (mulhi(n, m) + n) >> s for every 0 <= n < 2^31
static inline void k3x_fastdiv(int d, unsigned int& m, unsigned int& s) {
    s = 0;
    while ((1ll << s) < d) ++s;
    m = (unsigned int)((((1ull << (32 + s)) + (unsigned long long)d - 1) / (unsigned long long)d) - (1ull << 32));
}

template <int CK, int MT, int EPI, bool SUMS, bool HS, bool MULTI, bool FA = false, int YT = 4, bool EA = false>
static int k3x_launch_t(const G1Params& p_in, int tiles_total, int row_tiles, hipStream_t stream) {
    using GEO = K3XGeom<CK, MT, YT>;
    G1Params p = p_in;
    if (YT != 4) {                                       // re-tile the volume in 4 x YT x 16 tiles
        p.tyn = (p.H + YT - 1) / YT;
        p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
        tiles_total = p.tiles_per_sample * p.N;
    }
    if (FA && (p.N * p.C > 192 || !p.x_stats || !p.fa_sums)) return VS_ESHAPE;      // waves 1 .. 3 build the fused-apply tables
    const size_t tables = (size_t)(FA ? 6 : 2) * p.N * p.C * sizeof(float) + (p.sums ? (size_t)2 * p.N * p.M * sizeof(float) : 0);
    const size_t lds = K3X_LDS_TILE + (size_t)3 * GEO::PLANE_BYTES + GEO::W_BYTES + tables;
    if (lds > 160 * 1024) return VS_ESHAPE;
    // buffer offsets are 32-bit bytes, signed on the device
    if ((long long)p.N * p.D * p.H * p.W * p.C * 4 >= 2147483648ll || (long long)p.N * p.D * p.H * p.W * p.M * 4 >= 2147483648ll) return VS_ESHAPE;
    k3x_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k3x_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k3x_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    if (SUMS != (p.sums != nullptr) || (SUMS && !FA && p.x_stats != nullptr) || MULTI != (p.nch > 1)) return VS_EINVAL;
    auto kern = k3x_kernel<CK, MT, EPI, SUMS, HS, MULTI, FA, YT, EA>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    // persistent grid: as many workgroups as the LDS lets a CU hold (CK = 8: two, CK = 16: one), each walking a strided slice of the tile list
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    int wg = 256 * per_cu / (row_tiles < per_cu ? row_tiles : per_cu);
    if (wg < 256) wg = 256;
    if (row_tiles > per_cu) wg = (256 * per_cu + row_tiles - 1) / row_tiles / 8 * 8;
    if (wg < 8) wg = 8;
    const int gx = tiles_total < wg ? tiles_total : wg;
    if (EA) {                                            // one tile per workgroup, every workgroup resident while its sample's peers wait for it
        if (gx != tiles_total || (long long)tiles_total * row_tiles > k3x_ea_max_wgs(p.N, p.C, p.M) || !p.ea_sync || !p.ea_fault) return VS_ESHAPE;
        p.ea_items = p.tiles_per_sample * row_tiles;
    }
    hipLaunchKernelGGL(kern, dim3(gx, row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// the fused-apply instantiations: 8-channel chunks, 16-row workgroups (the 16 -> 16 layers of the 48^3 level and their 64^3 / 80^3 counterparts)
template <int CK, int MT, int EPI, bool MULTI>
constexpr bool k3x_has_fa() { return CK == 8 && MT == 16 && EPI == EPI_RAW && MULTI; }

// short tiles (YT = 2 / 1): the plain, lazy-input and fused-sums forms only
template <int CK, int MT, bool MULTI, int YT>
static int k3x_launch_short(const G1Params& p, int tiles_total, int row_tiles, hipStream_t stream) {
    if (p.fa_x != nullptr) return VS_ESHAPE;
    if (p.ea_sync != nullptr) {
        if (!p.sums) return VS_EINVAL;
        return k3x_launch_t<CK, MT, EPI_RAW, true, false, MULTI, false, YT, true>(p, tiles_total, row_tiles, stream);
    }
    if (p.sums != nullptr) return k3x_launch_t<CK, MT, EPI_RAW, true, false, MULTI, false, YT>(p, tiles_total, row_tiles, stream);
    if (p.x_stats != nullptr) return k3x_launch_t<CK, MT, EPI_RAW, false, true, MULTI, false, YT>(p, tiles_total, row_tiles, stream);
    return k3x_launch_t<CK, MT, EPI_RAW, false, false, MULTI, false, YT>(p, tiles_total, row_tiles, stream);
}

template <int CK, int MT, int EPI, bool MULTI>
static int k3x_launch(const G1Params& p, int tiles_total, int row_tiles, hipStream_t stream) {
    if (p.fa_x != nullptr) {
        if constexpr (k3x_has_fa<CK, MT, EPI, MULTI>()) {
            return p.sums ? k3x_launch_t<CK, MT, EPI, true, false, MULTI, true>(p, tiles_total, row_tiles, stream)
                          : k3x_launch_t<CK, MT, EPI, false, false, MULTI, true>(p, tiles_total, row_tiles, stream);
        } else return VS_ESHAPE;
    }
    if (p.sums != nullptr) {
        if constexpr (EPI == EPI_RAW) return k3x_launch_t<CK, MT, EPI, true, false, MULTI>(p, tiles_total, row_tiles, stream);
        else return VS_EINVAL;
    }
    if (p.x_stats != nullptr) return k3x_launch_t<CK, MT, EPI, false, true, MULTI>(p, tiles_total, row_tiles, stream);
    return k3x_launch_t<CK, MT, EPI, false, false, MULTI>(p, tiles_total, row_tiles, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// k3xt_kernel: the limb kernel for 8 stored input AND output channels (in_block, up5's second and third conv, out_block and their backward-data:
// the full-resolution layers, a third of the fp32 step).  With 8 real rows half of every MFMA of k3x_kernel multiplies padding, and those launches
// are bound by MFMA and LDS-read cycles (50-70 us at 96^3 against 25 us of HBM time).  As in k3t_kernel (igemm_k3t.h) the 16 MFMA rows become
// TOEPLITZ rows (dx2, co) — two x-adjacent output voxels times 8 channels — and a k-group is one (tz, ty) pair with k = (window position 0..3, ci):
// an MFMA column set covers 32 voxels of a row, 9 k-groups x 6 limb products per 32 voxels instead of 14 x 6, and every accumulator lane is a real
// output.  Tile 4 x 2 x 32 (halo 6 x 4 x 34: three limb planes 43 KB + weight block 27 KB -> two workgroups per CU); a wave owns one z-plane:
// two rows of 32 voxels.  Weights: the Toeplitz VS_F32X3 image (pack.hip), value W[co][ci][tz][ty][xpos - dx2] or 0.
#define K3XT_HX 34
#define K3XT_HY 4
#define K3XT_TV (6 * K3XT_HY * K3XT_HX)
#define K3XT_NIT ((K3XT_TV * 2 + 255) / 256)
#define K3XT_PB (K3XT_NIT * 256 * 8)
#define K3XT_NWF (9 * 3 * 64)
#define K3XT_NWI ((K3XT_NWF + 255) / 256)
#define K3XT_WB (K3XT_NWI * 256 * 16)

// FA (backward-data only; round 5): the input gradient arrives UN-applied, as in k3t_kernel's fused apply (igemm_k3t.h) — p.x = g = dL/da of the lazy
// activation a = relu(norm(p.fa_x)), with that activation's statistics (p.x_stats) and IN-backward sums (p.fa_sums) — and rstd * (g [xhat > 0] - m1 - xhat m2)
// is evaluated in fp32 on the staged fragments before the limb split; the applied gradient of the tile's centre voxels goes to p.fa_dx when given (the
// layer's weight gradient reads it).  The standalone apply pass of the parity mode is three passes over a 56 MB tensor at 96^3 (~40 us).
template <int EPI, bool SUMS, bool HS, bool FA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k3xt_kernel(const G1Params p) {
    static_assert(!FA || (!HS && EPI == EPI_RAW), "fused apply: backward-data use");
    constexpr int TV = K3XT_TV, PLANE = K3XT_HY * K3XT_HX, NIT = K3XT_NIT, PB = K3XT_PB, NWI = K3XT_NWI, NWF = K3XT_NWF, NU = TV * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3X_LDS_RED);
    char* s_tile = smem + K3X_LDS_TILE;
    char* s_w = s_tile + 3 * PB;
    float* s_mean = (float*)(s_w + K3XT_WB);
    float* s_rstd = s_mean + p.N * 8;
    float* s_mkm = s_rstd + p.N * 8;
    float* s_mkr = s_mkm + p.N * 8;
    float* s_fa = s_mkr + p.N * 8;                       // FA: rstd, -mean*rstd, m1, m2 of the input gradient's activation, [N*8] each

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * 4));
    const i32x4 frsrc = make_rsrc(FA ? p.fa_x : p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * 4));
    const i32x4 dxrsrc = make_rsrc(FA && p.fa_dx != nullptr ? p.fa_dx : p.y, (FA && p.fa_dx != nullptr) ? (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * 4) : 0u);
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    const double* st_src = SUMS ? p.mask_stats : p.x_stats;
    const int st_n = (SUMS || HS) ? p.N * 8 : 0;
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(st_src, (size_t)tid, (size_t)st_n, st_pre);
    double fa_pre[2][2] = {{0.0, 1.0}, {0.0, 0.0}};      // FA: wave 1 requests the activation's (sum, sumsq) and (sum g*mask, sum g*mask*xhat) pairs
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * 8) {
            stat_load(p.x_stats, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[0]);
            stat_load(p.fa_sums, (size_t)(tid - 64), (size_t)p.N * 8, fa_pre[1]);
        }
    }

    // fragment b = channels 4*part .. of halo voxel tv_b = (tid + 256 b) >> 1
    const int part = tid & 1;
    int rel_off[NIT], tzyx[NIT];
    unsigned int cbits = 0;                              // FA: fragment b belongs to a centre (non-halo) voxel of the 4 x 2 x 32 tile
    const int lds_w0 = (tid >> 1) * 16 + part * 8;        // + b * 2048, + limb * PB
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u >> 1;
        const int tx_ = tv % K3XT_HX, ty_ = (tv / K3XT_HX) % K3XT_HY, tz_ = tv / PLANE;
        rel_off[b] = (((tz_ * p.H + ty_) * p.W + tx_) * 8 + part * 4) * 4;
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;
        cbits |= (u < NU && tz_ >= 1 && tz_ <= 4 && ty_ >= 1 && ty_ <= 2 && tx_ >= 1 && tx_ <= 32) ? (1u << b) : 0u;
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int f = tid + i * 256;
        w_off[i] = f > NWF - 1 ? NWF - 1 : f;
    }
    u32x4 xv[NIT], wv[NWI], fv[FA ? NIT : 1];
    unsigned int okbits = 0;
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {
        Coord c;
        c.n = fdiv(t, p.fd_m[0], p.fd_s[0]);
        const int tl = t - c.n * p.tiles_per_sample;
        const int tz = fdiv(tl, p.fd_m[1], p.fd_s[1]);
        const int r = tl - tz * (p.txn * p.tyn);
        const int ty = fdiv(r, p.fd_m[2], p.fd_s[2]);
        c.z0 = tz * 4; c.y0 = ty * 2; c.x0 = (r - ty * p.txn) * 32;
        return c;
    };
    auto load_x = [&](const Coord& c) {
        const int base = (((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * 8 * 4;
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
    auto write_x_fa = [&](const Coord& c) {             // FA: apply pass in fp32 on the staged fragments, then the limb split [+ the applied gradient of the centre voxels]
        float rr[4], ss[4], aa[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rr[j] = s_fa[0 * p.N * 8 + c.n * 8 + part * 4 + j];
            ss[j] = s_fa[1 * p.N * 8 + c.n * 8 + part * 4 + j];
            aa[j] = s_fa[2 * p.N * 8 + c.n * 8 + part * 4 + j];
            bb[j] = s_fa[3 * p.N * 8 + c.n * 8 + part * 4 + j];
        }
        const int base = (((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * 8 * 4;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const bool ok = (okbits >> b) & 1u;           // out-of-volume halo voxels: the gradient is zero-padded
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gq = __uint_as_float(xv[b][j]);
                const float xh = __uint_as_float(fv[b][j]) * rr[j] + ss[j];
                const float gm = xh > 0.f ? gq : 0.f;
                const float d = rr[j] * (gm - aa[j] - xh * bb[j]);
                v[j] = ok ? d : 0.f;
            }
            unsigned int lm[3][2];
            vs_limb_split4(v, lm);
#pragma unroll
            for (int l = 0; l < 3; ++l) *(u32x2*)(s_tile + l * PB + lds_w0 + b * 2048) = u32x2{lm[l][0], lm[l][1]};
            if (p.fa_dx != nullptr)                        // workgroup-uniform
                vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), dxrsrc, (ok && ((cbits >> b) & 1u)) ? base + rel_off[b] : -1, 0, 0);
        }
    };
    auto write_x = [&](int n) {
        float mn[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (HS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mn[j] = s_mean[n * 8 + part * 4 + j]; rs[j] = s_rstd[n * 8 + part * 4 + j]; }
        }
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(xv[b][j]);
            if constexpr (HS) {
                const bool ok = (okbits >> b) & 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = (v[j] - mn[j]) * rs[j];
                    v[j] = ok ? fmaxf(t, 0.f) : 0.f;
                }
            }
            unsigned int lm[3][2];
            vs_limb_split4(v, lm);
#pragma unroll
            for (int l = 0; l < 3; ++l) *(u32x2*)(s_tile + l * PB + lds_w0 + b * 2048) = u32x2{lm[l][0], lm[l][1]};
        }
    };

    int t, t_end, G;
    if (((int)gridDim.x & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7;
        G = (int)gridDim.x >> 3;
        t = (int)(((long long)total_tiles * xcd) >> 3) + ((int)blockIdx.x >> 3);
        t_end = (int)(((long long)total_tiles * (xcd + 1)) >> 3);
    } else { G = (int)gridDim.x; t = (int)blockIdx.x; t_end = total_tiles; }
    Coord cur = tile_coord(t), nxt = cur;
#pragma unroll
    for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i]];
    load_x(cur);
    // this lane's accumulator rows 4g .. 4g+3 = (dx2 = g >> 1, channels 4 (g & 1) ..)
    const int dx2 = g >> 1, ch0 = 4 * (g & 1);
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (p.bias != nullptr && ch0 + r < (EPI == EPI_SOFTMAX2 ? 2 : p.M)) ? p.bias[ch0 + r] : 0.f;
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * 4));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * 8 * 4));
    if (tid < st_n) {
        float m, r;
        pair_to_mean_rstd(st_pre, SUMS ? p.inv_count_out : p.inv_count_in, p.eps, m, r);
        if constexpr (SUMS) { s_mkm[tid] = m; s_mkr[tid] = r; }
        else { s_mean[tid] = m; s_rstd[tid] = r; }
    }
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * 8) {
            const int i = tid - 64;
            float m, r;
            pair_to_mean_rstd(fa_pre[0], p.inv_count_in, p.eps, m, r);
            s_fa[0 * p.N * 8 + i] = r;
            s_fa[1 * p.N * 8 + i] = -m * r;
            s_fa[2 * p.N * 8 + i] = (float)(fa_pre[1][0] * p.inv_count_in);
            s_fa[3 * p.N * 8 + i] = (float)(fa_pre[1][1] * p.inv_count_in);
        }
    }
    // B fragment of (k-group (tz, ty), row cg, lane (col, g)): limb plane + ((wave + tz) * PLANE + (cg + ty) * HX + 2 col + g) * 16
    const int baddr = (wave * PLANE + 2 * col + g) * 16;
    const char* s_wl = s_w + lane * 16;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[r]));
#pragma unroll
    for (int i = 0; i < NWI; ++i) *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    bool first = true;
    __syncthreads();

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave, ox = x0 + 2 * col + dx2;
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + ox) * 8 + ch0) * 4;          // + cg * W * 8 * 4
        const bool zx_ok = oz < p.D && ox < p.W;
        f32x4 acc[2], acl[2];
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) { acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[cg] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        u32x4 mk[2];
        if (!first) __syncthreads();
        if constexpr (FA) write_x_fa(cur); else write_x(n);
        first = false;
        __syncthreads();
        if constexpr (EPI == EPI_RAW && SUMS) {
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const bool valid = zx_ok && y0 + cg < p.H && ch0 < p.M;
                mk[cg] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(mrsrc, valid ? ebase + cg * p.W * 32 : -1, 0, 0));
            }
        }
        nxt = tile_coord(t + G);
        if (t + G < t_end) load_x(nxt);
#pragma unroll
        for (int kg = 0; kg < 9; ++kg) {
            const int tz = kg / 3, ty = kg % 3;
            u32x4 a[3], b[2][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) a[j] = *(const u32x4*)(s_wl + (kg * 3 + j) * 1024);
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                for (int i = 0; i < 3; ++i) b[cg][i] = *(const u32x4*)(s_tile + i * PB + baddr + (tz * PLANE + (cg + ty) * K3XT_HX) * 16);
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                acl[cg] = mfma16(a[0], b[cg][2], acl[cg], (unsigned short*)nullptr);
                acl[cg] = mfma16(a[1], b[cg][1], acl[cg], (unsigned short*)nullptr);
                acl[cg] = mfma16(a[2], b[cg][0], acl[cg], (unsigned short*)nullptr);
                acl[cg] = mfma16(a[0], b[cg][1], acl[cg], (unsigned short*)nullptr);
                acl[cg] = mfma16(a[1], b[cg][0], acl[cg], (unsigned short*)nullptr);
                acc[cg] = mfma16(a[0], b[cg][0], acc[cg], (unsigned short*)nullptr);
            }
        }
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) acc[cg] += acl[cg];

        if constexpr (EPI == EPI_SOFTMAX2) {
            if ((g & 1) == 0) {                              // the lanes holding channels 0 .. 3 of voxel ox: the two logits are channels 0 and 1
                const size_t V = (size_t)p.D * p.H * p.W;
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    const int oy = y0 + cg;
                    if (!(oz < p.D && oy < p.H && ox < p.W)) continue;
                    float l0 = acc[cg][0] + bv[0], l1 = acc[cg][1] + bv[1];
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
            const bool rvalid = ch0 < p.M;
            float mm[4] = {0.f, 0.f, 0.f, 0.f}, mr[4] = {0.f, 0.f, 0.f, 0.f};
            if (SUMS && rvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * 8 + ch0 + r]; mr[r] = s_mkr[n * 8 + ch0 + r]; }
            }
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const bool valid = rvalid && zx_ok && y0 + cg < p.H;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[cg][r] + bv[r];
                vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), yrsrc, valid ? ebase + cg * p.W * 32 : -1, 0, 0);
                if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                if constexpr (SUMS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (__uint_as_float(mk[cg][r]) - mm[r]) * mr[r];
                        const float gm = xh > 0.f ? v[r] : 0.f;
                        ssum[r] += gm; ssq[r] += gm * xh;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[r] += v[r]; ssq[r] += v[r] * v[r]; }
                }
            }
            double* const red_dst = SUMS ? p.sums : p.y_stats;
            if (red_dst != nullptr) {
                const bool flush = t + G >= t_end || nxt.n != n;       // workgroup-uniform
                if (flush) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = row16_sum(ssum[r]), q = row16_sum(ssq[r]);
                        if (col == 0) {
                            s_red[(wave * 16 + 4 * g + r) * 2 + 0] = s;
                            s_red[(wave * 16 + 4 * g + r) * 2 + 1] = q;
                        }
                        ssum[r] = 0.f; ssq[r] = 0.f;
                    }
                    __syncthreads();
                    if (tid < 16) {                              // channel tid >> 1: rows (dx2 = 0, ch) and (dx2 = 1, ch) = accumulator rows ch and ch + 8, of the four waves
                        const int chn = tid >> 1, st = tid & 1;
                        if (chn < p.M) {
                            double tot = 0.0;
#pragma unroll
                            for (int w = 0; w < 4; ++w) tot += (double)s_red[(w * 16 + chn) * 2 + st] + (double)s_red[(w * 16 + chn + 8) * 2 + st];
                            stat_add(red_dst, (size_t)n * p.M + chn, (size_t)p.N * p.M, st, tot);
                        }
                    }
                    if (t + G < t_end) __syncthreads();
                }
            }
        }
        cur = nxt;
    }
}

template <int EPI, bool SUMS, bool HS, bool FA = false>
static int k3xt_launch_t(const G1Params& p_in, hipStream_t stream) {
    G1Params p = p_in;
    p.tyn = (p.H + 1) / 2; p.txn = (p.W + 31) / 32;
    p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
    const int tiles_total = p.tiles_per_sample * p.N;
    const size_t lds = K3X_LDS_TILE + (size_t)3 * K3XT_PB + K3XT_WB + (size_t)(FA ? 8 : 4) * p.N * 8 * sizeof(float);
    if (lds > 160 * 1024 || p.N * 8 > (FA ? 192 : 256)) return VS_ESHAPE;          // FA: waves 1 .. 3 build the fused-apply tables
    if (FA && (!p.x_stats || !p.fa_x || !p.fa_sums)) return VS_EINVAL;
    if ((long long)p.N * p.D * p.H * p.W * 8 * 4 >= 2147483648ll) return VS_ESHAPE;
    k3x_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k3x_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k3x_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    if (SUMS != (p.sums != nullptr) || (SUMS && !FA && p.x_stats != nullptr) || p.C != 8 || p.M != 8) return VS_EINVAL;
    auto kern = k3xt_kernel<EPI, SUMS, HS, FA>;
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    const int wg = 512;                                  // two workgroups per CU (73 KB of LDS each)
    const int gx = tiles_total < wg ? tiles_total : wg;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <int EPI>
static int k3xt_launch(const G1Params& p, hipStream_t stream) {
    if (p.fa_x != nullptr) {                              // un-applied gradient in: never through a kernel that ignores fa_x
        if constexpr (EPI == EPI_RAW) return p.sums != nullptr ? k3xt_launch_t<EPI, true, false, true>(p, stream) : k3xt_launch_t<EPI, false, false, true>(p, stream);
        else return VS_EINVAL;
    }
    if (p.sums != nullptr) {
        if constexpr (EPI == EPI_RAW) return k3xt_launch_t<EPI, true, false>(p, stream);
        else return VS_EINVAL;
    }
    if (p.x_stats != nullptr) return k3xt_launch_t<EPI, false, true>(p, stream);
    return k3xt_launch_t<EPI, false, false>(p, stream);
}
