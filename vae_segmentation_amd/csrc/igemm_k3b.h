// 3x3x3 (pad 1) implicit-GEMM convolution, bf16 or fp16 operands / fp32 accumulate: forward and backward-data.
//
// Second-generation kernel for the bf16 path (the fp32 path keeps k3_kernel, igemm_k3.h).  Phase stamps of k3_kernel
// (tools/stamps_k3.py) showed that no phase was bound by MFMA or HBM; every phase paid a memory round trip instead:
//   * weight fragments came straight from global inside the MFMA loop, two loads in flight, queued (vmcnt retires in
//     issue order) behind the next tile's activation prefetch;
//   * bounds-tested staging compiled to exec-mask branches, whose joins made the compiler drain vmcnt(0) before every
//     epilogue store and inside the prefetch itself;
//   * normalise-on-load cost ~45 VALU per 16-byte fragment, repeated by every row-block workgroup over the 2.5x halo.
// Here a stage = (tile, channel chunk) is fetched whole into registers one stage ahead — activations through
// bounds-checked *buffer* loads (out-of-volume fragments read as zero, no branches), weights as the workgroup's
// [row block][k-group][lane] fragment block, loaded cooperatively once instead of once per wave — and both go through LDS.
// The MFMA loop then reads A and B from LDS with immediate offsets only (no VALU, no vmcnt), the halo tile of the 32-channel
// chunks is XOR-swizzled so every ds_read_b128 lane group covers all 16 slots of a bank row, and normalise+ReLU runs as
// packed fp32 fma / packed bf16 max (~24 VALU per fragment).  Epilogues, statistics and fragment formats are those of
// k3_kernel.
#pragma once
#include <stdlib.h>
#include "igemm.h"
#include "chain.h"

#ifdef VS_STAMPS   // diagnostic build only (tools/build_stamps.sh, tools/stamps_k3.py): per-phase cycle sums of wave 0
__device__ unsigned long long g_k3_stamps[2048 * 8];
extern "C" int vs_debug_read_k3_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k3_stamps), sizeof(unsigned long long) * n);
}
#define K3_TICK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk_acc[i] += now_ - tk_last; tk_last = now_; } while (0)
#define K3_TICK_INIT unsigned long long tk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk_last = __builtin_amdgcn_s_memtime(); tk_acc[7] = __builtin_amdgcn_s_memrealtime();
#define K3_TICK_FLUSH do { if (threadIdx.x == 0) { tk_acc[7] = (tk_acc[7] << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull); for (int i_ = 0; i_ < 8; ++i_) g_k3_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 2047) * 8 + i_] = tk_acc[i_]; } } while (0)
#elif defined(VS_STAMPS_LITE)   // only the workgroup's start / end on the 100 MHz clock (no extra VGPRs: the occupancy of the release build)
__device__ unsigned long long g_k3_stamps[2048 * 8];
extern "C" int vs_debug_read_k3_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k3_stamps), sizeof(unsigned long long) * n);
}
#define K3_TICK(i)
#define K3_TICK_INIT const unsigned long long tk_rt0 = __builtin_amdgcn_s_memrealtime();
#define K3_TICK_FLUSH do { if (threadIdx.x == 0) { unsigned long long* d_ = g_k3_stamps + ((blockIdx.y * gridDim.x + blockIdx.x) & 2047) * 8; d_[0] = 1; d_[7] = (tk_rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull); } } while (0)
#else
#define K3_TICK(i)
#define K3_TICK_INIT
#define K3_TICK_FLUSH
#endif

#define K3B_LDS_RED 0          // float[4][64][2]
#define K3B_LDS_TAPS 2048      // int[64]: byte offset of (k-group, lane group)'s tap in the halo tile (C = 8 / 16)
#define K3B_LDS_TILE 2304      // halo tile, weight block, then the per-(n,c) tables

// YT = tile extent in y (4 or 8): a wave owns YT rows of 16 voxels of its z-plane.  YT = 8 (halo 6x10x18: 2.1x the outputs
// instead of 2.5x, half the per-tile bookkeeping) is used for the large-volume small-channel layers.
template <int CK, int MT, int YT = 4>
struct K3BGeom {
    static constexpr int RB = MT / 16;
    static constexpr int TV = 6 * (YT + 2) * 18;                             // staged halo voxels
    static constexpr bool SMALLC = CK < 32;                                  // several taps per 32-wide k-group, one chunk
    static constexpr int NKGC = SMALLC ? (27 * CK + 31) / 32 : 27;           // k-groups per channel chunk
    static constexpr int CKB = CK * 2;
    static constexpr int TILE_BYTES = ((TV * (CKB / 16) + 255) / 256) * 256 * 16;     // padded: every thread stores all its fragments
    static constexpr int NWF = RB * NKGC * 64;                               // 16-byte weight fragments per chunk per workgroup
    static constexpr int W_BYTES = ((NWF + 255) / 256) * 256 * 16;                     // padded likewise
};

// register budget: 4 workgroups per CU (one wave per SIMD each) for the small-channel kernels, 2 for the 32-channel chunks
// SUMS: backward-data use (input = a materialised gradient, no statistics; epilogue accumulates the fused IN-backward sums)
// HS: the input is a lazy activation — compile-time, like every condition on the staging path (a run-time test between a load and its
// use, or an exec-masked tail store, makes the compiler drain vmcnt(0): it then waits for the prefetched stage as well)
// T: unsigned short (bf16 bits) or vs_half (fp16) — same fragment shapes, same MFMA rate; last template argument so that the profiler's
// kernel names keep their prefix
// FA (backward-data use; CK == 32: any chunk count, one wave per SIMD — the second staged operand does not fit 256 VGPRs — the 24^3 / 12^3 levels):
// the input gradient arrives un-applied (p.x = g = dL/da of a = relu(norm(p.fa_x)), statistics p.x_stats,
// IN-backward sums p.fa_sums) and the apply pass runs while the halo tile is staged; centre voxels also go to p.fa_dx when given (see
// igemm_k3t.h, where the same is done for the 8 -> 8 layers)
// EA (backward-data with fused sums, ONE tile per workgroup, every workgroup of the launch resident — the 24^3 / 12^3 levels; round 6): the epilogue keeps its
// rounded outputs and the mask values in registers, adds its partial sums, arrives on its SAMPLE's counter (chain.h's hand-off: InstanceNorm's dependency
// domain is the sample), waits for the sample's other workgroups, reads the complete sums back (sc1) and stores the APPLIED gradient
// rstd * (g * [xhat > 0] - m1 - xhat * m2) — in_relu_bwd_apply_kernel's arithmetic on the same rounded values, bit for bit in the deterministic build.
// The un-applied tensor is never written, the standalone apply launch (6.2 us for ~1 us of work at these sizes) disappears.
template <int CK, int MT, int EPI, bool SUMS, int YT = 4, bool HS = false, typename T = unsigned short, bool FA = false, bool EA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(
    YT == 8 ? 2 : (CK == 32 ? ((MT == 32 || FA) ? 1 : 2) : (MT == 32 ? (SUMS ? 2 : 3) : ((CK == 16 && SUMS) || FA ? 3 : 4))), 8))) void k3b_kernel(const G1Params p) {
    K3_TICK_INIT
    using GEO = K3BGeom<CK, MT, YT>;
    static_assert(!FA || (!HS && EPI == EPI_RAW), "fused apply: backward-data kernels");
    static_assert(!EA || (SUMS && !FA && !HS && EPI == EPI_RAW), "epilogue apply: backward-data kernels with fused sums");
    static_assert(YT == 4 || (YT == 8 && CK < 32 && MT == 16) || ((YT == 2 || YT == 1) && CK == 32 && MT == 16), "tall tiles: single-chunk layers, 16 rows; short tiles: 32-channel chunks, 16 rows");
    constexpr int TV = GEO::TV, PLANE = (YT + 2) * 18;
    static_assert(CK == 8 || CK == 16 || CK == 32, "chunk width");
    constexpr int RB = GEO::RB, NKGC = GEO::NKGC, CKB = GEO::CKB, NWF = GEO::NWF;
    constexpr bool SMALLC = GEO::SMALLC;
    constexpr int U = CKB / 16;                          // 16-byte fragments per staged voxel
    constexpr int NU = TV * U;
    constexpr int NIT = (NU + 255) / 256;                // activation fragments per thread per stage
    constexpr int NWI = (NWF + 255) / 256;               // weight fragments per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + K3B_LDS_RED);
    int* s_taps = (int*)(smem + K3B_LDS_TAPS);
    char* s_tile = smem + K3B_LDS_TILE;
    char* s_w = s_tile + GEO::TILE_BYTES;
    float* s_scale = (float*)(s_w + GEO::W_BYTES);       // rstd and -mean*rstd of the lazy input, [N*C] each
    float* s_shift = s_scale + p.N * p.C;
    float* s_mkm = s_shift + p.N * p.C;                  // mean / rstd of the mask tensor's channels (fused IN-bwd sums)
    float* s_mkr = s_mkm + p.N * p.M;
    float* s_fa = s_shift + p.N * p.C + (SUMS ? 2 * p.N * p.M : 0);   // FA: rstd, -mean*rstd, m1, m2 of the input gradient's activation, [N*C] each

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * RB;
    constexpr bool has_stats = HS;
    constexpr bool has_sums = SUMS;
    const int total_tiles = p.tiles_per_sample * p.N;
    const i32x4 xrsrc = make_rsrc(p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 2));
    const i32x4 frsrc = make_rsrc(FA ? p.fa_x : p.x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 2));
    const i32x4 dxrsrc = make_rsrc(FA && p.fa_dx != nullptr ? p.fa_dx : p.x, (FA && p.fa_dx != nullptr) ? (unsigned int)((long long)p.N * p.D * p.H * p.W * p.C * 2) : 0u);
    const u32x4* __restrict__ wp = (const u32x4*)p.wp;

    // the (sum, sumsq) pair this thread turns into a table entry is requested first of all: it is the oldest load in the
    // queue (vmcnt retires in order), so the tables are built while the first stage's fragments are still in flight
    const double* st_src = SUMS ? p.mask_stats : p.x_stats;
    const int st_n = SUMS ? p.N * p.M : (has_stats ? p.N * p.C : 0);
    double st_pre[2] = {0.0, 1.0};
    if (tid < st_n) stat_load(st_src, (size_t)tid, (size_t)st_n, st_pre);
    double fa_pre[2][2] = {{0.0, 1.0}, {0.0, 0.0}};      // FA: waves 1.. request the activation's (sum, sumsq) and (sum g*mask, sum g*mask*xhat) pairs
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * p.C) {
            stat_load(p.x_stats, (size_t)(tid - 64), (size_t)p.N * p.C, fa_pre[0]);
            stat_load(p.fa_sums, (size_t)(tid - 64), (size_t)p.N * p.C, fa_pre[1]);
        }
    }

    // ---- per-thread stage geometry (tile independent) -----------------------------------------------------------------
    // fragment b of this thread is 16-byte part `part` (the same for every b: 256 % U == 0) of tile voxel tv_b
    const int part = tid % U;
    // LDS byte address of fragment b = lds_w0 + b * 4096 (tv advances by 256 / U voxels of CKB bytes per b), with the
    // 32-channel tile's swizzle bit of fragment b kept in swzbits
    int rel_off[NIT], tzyx[NIT];
    const int lds_w0 = (tid / U) * CKB;
    unsigned int swzbits = 0, cbits = 0;                  // cbits (FA): fragment b belongs to a centre (non-halo) voxel of the tile
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int u = tid + b * 256;
        const int tv = u / U;
        const int tx_ = tv % 18, ty_ = (tv / 18) % (YT + 2), tz_ = tv / PLANE;
        cbits |= (u < NU && tz_ >= 1 && tz_ <= 4 && ty_ >= 1 && ty_ <= YT && tx_ >= 1 && tx_ <= 16) ? (1u << b) : 0u;
        rel_off[b] = (((tz_ * p.H + ty_) * p.W + tx_) * p.C + part * 8) * 2;              // bytes from the tile's (0,0,0) halo voxel
        tzyx[b] = u < NU ? (tz_ | (ty_ << 8) | (tx_ << 16)) : 0x00ffffff;                 // out-of-list fragments fail every bounds test
        if (CK == 32) swzbits |= (unsigned int)((tx_ >> 2) & 1) << b;                     // see baddr[] below
    }
    int w_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int f = tid + i * 256;
        if (f > NWF - 1) f = NWF - 1;
        const int rb = f / (NKGC * 64), r = f - rb * (NKGC * 64);
        w_off[i] = (rb0 + rb) * (p.nch * NKGC * 64) + r;                                   // + ch * NKGC * 64
    }

    u32x4 xv[NIT], wv[NWI], fv[FA ? NIT : 1];
    unsigned int okbits = 0;
    struct Coord { int n, z0, y0, x0; };
    auto tile_coord = [&](int t) {                        // scalar: t is workgroup-uniform
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
        for (int i = 0; i < NWI; ++i) wv[i] = wp[w_off[i] + ch * (NKGC * 64)];
    };
    auto load_x = [&](const Coord& c, int ch) {
        const int n = c.n, z0 = c.z0, y0 = c.y0, x0 = c.x0;
        const int base = ((((n * p.D + z0 - 1) * p.H + y0 - 1) * p.W + x0 - 1) * p.C + ch * CK) * 2;
        okbits = 0;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int gz = z0 - 1 + (tzyx[b] & 0xff), gy = y0 - 1 + ((tzyx[b] >> 8) & 0xff), gx = x0 - 1 + (tzyx[b] >> 16);
            const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            okbits |= ok ? (1u << b) : 0u;
            xv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, ok ? base + rel_off[b] : -1, 0, 0));
            if constexpr (FA) fv[b] = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(frsrc, ok ? base + rel_off[b] : -1, 0, 0));
        }
    };
    auto write_x_fa = [&](const Coord& c, int ch) {     // FA: apply pass on the staged fragments [+ the applied gradient of the centre voxels to fa_dx]
        f32x2 r2[4], s2[4], a2[4], b2[4];
        const int c0 = c.n * p.C + ch * CK + part * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r2[i] = *(const f32x2*)(s_fa + 0 * p.N * p.C + c0 + 2 * i);
            s2[i] = *(const f32x2*)(s_fa + 1 * p.N * p.C + c0 + 2 * i);
            a2[i] = *(const f32x2*)(s_fa + 2 * p.N * p.C + c0 + 2 * i);
            b2[i] = *(const f32x2*)(s_fa + 3 * p.N * p.C + c0 + 2 * i);
        }
        const int base = ((((c.n * p.D + c.z0 - 1) * p.H + c.y0 - 1) * p.W + c.x0 - 1) * p.C + ch * CK) * 2;
        // the applied gradient goes out once per tile: with several row-block workgroups per tile (gridDim.y > 1) only the first one stores it
        const bool store_dx = p.fa_dx != nullptr && blockIdx.y == 0;          // workgroup-uniform
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
            const int pw = CK == 32 ? (part ^ (int)(((swzbits >> b) & 1u) << 1)) : part;
            *(u32x4*)(s_tile + lds_w0 + b * 4096 + pw * 16) = v;
            if (store_dx)
                vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), dxrsrc, (ok && ((cbits >> b) & 1u)) ? base + rel_off[b] : -1, 0, 0);
        }
    };
    auto write_x = [&](int n, int ch) {
        f32x2 sc[4], sh[4];
        if (has_stats) {
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
            if (has_stats) {
                const u32x4 a = act8<T>(v, sc, sh);
                const bool ok = (okbits >> b) & 1u;       // zero padding applies to the normalised activation
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? a[i] : 0u;
            }
            const int pw = CK == 32 ? (part ^ (int)(((swzbits >> b) & 1u) << 1)) : part;
            *(u32x4*)(s_tile + lds_w0 + b * 4096 + pw * 16) = v;      // fragments beyond the tile are zeros in the padded tail
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            *(u32x4*)(s_w + (tid + i * 256) * 16) = wv[i];
    };

    // ---- first stage in flight before anything else; the statistics tables meanwhile ----------------------------------
    // XCD-aware walk: consecutive workgroup ids land on different XCDs (8, each with its own L2); the workgroups of one XCD work
    // on one contiguous run of tiles and find their neighbours' halos in that XCD's L2 (identity walk when the grid is not a
    // multiple of 8).
    // XCD x owns the contiguous run [x*T/8, (x+1)*T/8) of the tile list and deals it round-robin to its workgroups: every XCD gets
    // the same number of tiles (striding the whole list by the grid left the remainder T mod G to XCD 0 and 1).
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
    // bias of this workgroup's rows (registers: a load inside the tile loop would queue behind the prefetch)
    float bv[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (rb0 + rb) * 16 + 4 * g + r;
            bv[rb][r] = (p.bias != nullptr && row < (EPI == EPI_SOFTMAX2 ? 2 : p.M)) ? p.bias[row] : 0.f;
        }
    const i32x4 yrsrc = make_rsrc(p.y, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 2));
    const i32x4 mrsrc = make_rsrc(p.mask_x, (unsigned int)((long long)p.N * p.D * p.H * p.W * p.M * 2));
    for (int i = tid; i < st_n; i += 256) {
        double st[2] = {st_pre[0], st_pre[1]};
        if (i != tid) stat_load(st_src, (size_t)i, (size_t)st_n, st);
        float m, r;
        stats_to_mean_rstd_fast(st, SUMS ? p.inv_count_out : p.inv_count_in, p.eps, m, r);
        if constexpr (SUMS) { s_mkm[i] = m; s_mkr[i] = r; }
        else { s_scale[i] = r; s_shift[i] = -m * r; }
    }
    if constexpr (FA) {
        if (tid >= 64 && tid < 64 + p.N * p.C) {
            const int i = tid - 64;
            float m, r;
            stats_to_mean_rstd_fast(fa_pre[0], p.inv_count_in, p.eps, m, r);
            s_fa[0 * p.N * p.C + i] = r;
            s_fa[1 * p.N * p.C + i] = -m * r;
            s_fa[2 * p.N * p.C + i] = (float)(fa_pre[1][0] * p.inv_count_in);
            s_fa[3 * p.N * p.C + i] = (float)(fa_pre[1][1] * p.inv_count_in);
        }
    }

    // ---- LDS read addresses ---------------------------------------------------------------------------------------------
    // B fragment of (tap, column voxel (wave, cg, col)): tile voxel (wave + dz, cg + dy, col + dx).
    //  CK == 32: a lane reads 16-byte part g of the voxel; parts are stored XOR ((tx >> 2) & 1) << 1, which puts the 16 lanes
    //            of every ds_read_b128 lane group on 16 different slots of the 256-byte bank row for dx = 0, 1, 2
    //            (unswizzled: 2-way conflicts on every read).  Per-lane base per dx, everything else an immediate.
    //  CK < 32 : the lane's tap depends on g; per-lane byte offset per k-group, cg is an immediate.
    int baddr[3];
    if constexpr (SMALLC) {
        if (tid < NKGC * 4) {
            int tap = (tid >> 2) * (32 / CK) + ((tid & 3) * 8) / CK;
            if (tap > 26) tap = 13;                      // padded taps read the centre voxel (their weights are zero)
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            s_taps[tid] = (dz * PLANE + dy * 18 + dx) * CKB + (((tid & 3) * 8) % CK) * 2;
        }
        baddr[0] = (wave * PLANE + col) * CKB;
    } else {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) baddr[dx] = (wave * PLANE + col) * CKB + ((g ^ ((((col + dx) >> 2) & 1) << 1)) * 16);
    }
    const char* s_wl = s_w + lane * 16;

    float ssum[RB][4], ssq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; }

    // consume the bias here: its wait belongs to the prologue (left to the first use, the compiler waits for it inside
    // the tile loop, i.e. for the whole prefetch queued behind it, on every tile)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[rb][r]));
    constexpr bool restage_w = !SMALLC;                  // C = 8 / 16 is a single chunk: the weight block is staged once
    bool first = true;
    if constexpr (SMALLC) write_w();
    __syncthreads();                                     // tables visible
    K3_TICK(0);                                          // prologue

    for (; t < t_end; t += G) {
        const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
        const int oz = z0 + wave;
        // byte offset of output voxel (n, oz, y0 + cg, x0 + col), row 4g of row block rb: ebase + cg * W*M*2 + rb * 32; -1 = dropped
        const int ebase = ((((n * p.D + oz) * p.H + y0) * p.W + x0 + col) * p.M + rb0 * 16 + 4 * g) * 2;
        const bool zx_ok = oz < p.D && x0 + col < p.W;
        f32x4 acc[RB][YT];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < YT; ++cg) acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 mk[RB][YT];                                // mask tensor values under this tile's outputs (fused IN-bwd sums)
        i32x2 pkk[EA ? RB : 1][EA ? YT : 1];             // EA: this tile's rounded outputs, kept for the apply

        for (int ch = 0; ch < p.nch; ++ch) {
            if (!first) __syncthreads();                 // every wave is done reading the previous stage
            K3_TICK(1);
            if constexpr (FA) write_x_fa(cur, ch); else write_x(n, ch);
            if constexpr (restage_w) write_w();
            first = false;
            K3_TICK(2);
            __syncthreads();
            K3_TICK(3);
            // requests of the next stage, oldest-needed first (vmcnt retires in issue order)
            const bool last_ch = ch + 1 == p.nch;
            if constexpr (EPI == EPI_RAW) {
                if (has_sums && last_ch) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cg = 0; cg < YT; ++cg) {
                            const bool valid = zx_ok && y0 + cg < p.H && (rb0 + rb) * 16 + 4 * g < p.M;
                            mk[rb][cg] = __builtin_bit_cast(u32x2, vs_raw_buffer_load_b64(mrsrc, valid ? ebase + cg * p.W * p.M * 2 + rb * 32 : -1, 0, 0));
                        }
                }
            }
            {
                const int tn = last_ch ? t + G : t;
                if (last_ch) nxt = tile_coord(tn);
                if (tn < t_end) {
                    if constexpr (restage_w) load_w(last_ch ? 0 : ch + 1);
                    load_x(last_ch ? nxt : cur, last_ch ? 0 : ch + 1);
                }
            }

            K3_TICK(4);
            // ---- multiply this stage out of LDS ----
            // fragments of k-group kg+1 are read while k-group kg is multiplied; the scheduling barriers keep the compiler
            // from hoisting the whole unrolled loop's reads (hundreds of live registers) in front of the first MFMA
            auto read_kg = [&](int kg, u32x4 (&a)[RB], u32x4 (&b)[YT]) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = *(const u32x4*)(s_wl + (rb * NKGC + kg) * 1024);
                if constexpr (SMALLC) {
                    const int o = baddr[0] + s_taps[kg * 4 + g];
#pragma unroll
                    for (int cg = 0; cg < YT; ++cg) b[cg] = *(const u32x4*)(s_tile + o + cg * 18 * CKB);
                } else {
                    const int dz = kg / 9, dy = (kg / 3) % 3, dx = kg % 3;
#pragma unroll
                    for (int cg = 0; cg < YT; ++cg) b[cg] = *(const u32x4*)(s_tile + baddr[dx] + ((dz * PLANE + (dy + cg) * 18 + dx) * CKB));
                }
            };
            u32x4 fa[2][RB], fb[2][YT];
            read_kg(0, fa[0], fb[0]);
#pragma unroll
            for (int kg = 0; kg < NKGC; ++kg) {
                if (kg + 1 < NKGC) read_kg(kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < YT; ++cg) acc[rb][cg] = mfma16(fa[kg & 1][rb], fb[kg & 1][cg], acc[rb][cg], (T*)nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of this tile ----
        K3_TICK(5);
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
                if (has_sums && rvalid) {
                    const int row = (rb0 + rb) * 16 + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { mm[r] = s_mkm[n * p.M + row + r]; mr[r] = s_mkr[n * p.M + row + r]; }
                }
#pragma unroll
                for (int cg = 0; cg < YT; ++cg) {
                    const bool valid = rvalid && zx_ok && y0 + cg < p.H;
                    // round once to T; the statistics are those of the stored values
                    f32x2 lo, hi;
                    lo[0] = acc[rb][cg][0] + bv[rb][0]; lo[1] = acc[rb][cg][1] + bv[rb][1];
                    hi[0] = acc[rb][cg][2] + bv[rb][2]; hi[1] = acc[rb][cg][3] + bv[rb][3];
                    i32x2 pk;
                    pk[0] = (int)H16<T>::pack2(lo);
                    pk[1] = (int)H16<T>::pack2(hi);
                    if constexpr (EA) pkk[rb][cg] = pk;
                    else vs_raw_buffer_store_b64(pk, yrsrc, valid ? ebase + cg * p.W * p.M * 2 + rb * 32 : -1, 0, 0);
                    float v[4];
                    v[0] = H16<T>::lo((unsigned int)pk[0]); v[1] = H16<T>::hi((unsigned int)pk[0]);
                    v[2] = H16<T>::lo((unsigned int)pk[1]); v[3] = H16<T>::hi((unsigned int)pk[1]);
                    if (!valid) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                    if (has_sums) {
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
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[rb][r] += v[r]; ssq[rb][r] += v[r] * v[r]; }
                    }
                }
            }
            double* const red_dst0 = has_sums ? p.sums : p.y_stats;
            double* const red_dst = red_dst0;
            if (red_dst != nullptr) {
                const bool flush = t + G >= t_end || nxt.n != n;       // workgroup-uniform
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
                            stat_add(red_dst, (size_t)n * p.M + row, (size_t)p.N * p.M, st, tot);
                        }
                    }
                    if (t + G < t_end) __syncthreads();     // s_red is reused by a later flush
                }
            }
            if constexpr (EA) {
                // ---- the sample's sums are complete once all of its workgroups have arrived; then the apply, on registers ----
                unsigned int* ctr = p.ea_sync + (size_t)n * 256;          // 8 shards of 128 bytes per sample
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
                        const unsigned int gw[2] = {(unsigned int)pkk[rb][cg][0], (unsigned int)pkk[rb][cg][1]};
                        const u32x2 xx = mk[rb][cg];
                        const float gv[4] = {H16<T>::lo(gw[0]), H16<T>::hi(gw[0]), H16<T>::lo(gw[1]), H16<T>::hi(gw[1])};
                        const float xq[4] = {H16<T>::lo(xx[0]), H16<T>::hi(xx[0]), H16<T>::lo(xx[1]), H16<T>::hi(xx[1])};
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = vs_in_bwd_apply1(gv[r], xq[r], tb[r][0], tb[r][1], tb[r][2], tb[r][3]);
                        }
                        i32x2 po;
                        po[0] = (int)H16<T>::pack2(f32x2{o[0], o[1]});
                        po[1] = (int)H16<T>::pack2(f32x2{o[2], o[3]});
                        vs_raw_buffer_store_b64(po, yrsrc, valid ? ebase + cg * p.W * p.M * 2 + rb * 32 : -1, 0, 0);
                    }
                }
            }
        }
        cur = nxt;
        K3_TICK(6);
    }
    K3_TICK_FLUSH;
}

// k3b_kernel<32, 16, .., SUMS, .., EA>: how many workgroups of one launch are certainly resident together — two per CU (two waves per SIMD by the register
// budget) when two workgroups' LDS fit with room for the allocation granule, else one.  (Found the hard way: 10^3 x 256 rows needs 80 KB, one per CU; 288
// workgroups then waited for 32 that could not start, gave up after the bounded spin and applied incomplete sums.)
static inline int k3b_ea_max_wgs(int n, int c, int m) {
    using GEO = K3BGeom<32, 16, 4>;
    const size_t lds = K3B_LDS_TILE + (size_t)GEO::TILE_BYTES + GEO::W_BYTES + (size_t)2 * n * c * sizeof(float) + (size_t)2 * n * m * sizeof(float);
    return 2 * (lds + 1024) <= 160 * 1024 ? 512 : 256;
}

// (m, s) with n / d == (mulhi(n, m) + n) >> s for every 0 <= n < 2^31
static inline void k3b_fastdiv(int d, unsigned int& m, unsigned int& s) {
    s = 0;
    while ((1ll << s) < d) ++s;
    m = (unsigned int)((((1ull << (32 + s)) + (unsigned long long)d - 1) / (unsigned long long)d) - (1ull << 32));
}

template <typename T, int CK, int MT, int EPI, bool SUMS, int YT, bool HS, bool FA = false, bool EA = false>
static int k3b_launch_t(const G1Params& p_in, int tiles_total, int row_tiles, hipStream_t stream) {
    using GEO = K3BGeom<CK, MT, YT>;
    if (FA && (p_in.N * p_in.C > 192 || (CK < 32 && p_in.nch != 1) || !p_in.x_stats || !p_in.fa_sums)) return VS_ESHAPE;
    const size_t tables = (size_t)(FA ? 6 : 2) * p_in.N * p_in.C * sizeof(float) + (p_in.sums ? (size_t)2 * p_in.N * p_in.M * sizeof(float) : 0);
    const size_t lds = K3B_LDS_TILE + (size_t)GEO::TILE_BYTES + GEO::W_BYTES + tables;
    if (lds > 160 * 1024) return VS_ESHAPE;
    G1Params p = p_in;
    if (YT != 4) {                                       // re-tile the volume in 4 x YT x 16 tiles
        p.tyn = (p.H + YT - 1) / YT;
        p.tiles_per_sample = ((p.D + 3) / 4) * p.tyn * p.txn;
        tiles_total = p.tiles_per_sample * p.N;
    }
    // buffer offsets are 32-bit bytes, signed on the device
    if ((long long)p.N * p.D * p.H * p.W * p.C * 2 >= 2147483648ll || (long long)p.N * p.D * p.H * p.W * p.M * 2 >= 2147483648ll) return VS_ESHAPE;
    k3b_fastdiv(p.tiles_per_sample, p.fd_m[0], p.fd_s[0]);
    k3b_fastdiv(p.txn * p.tyn, p.fd_m[1], p.fd_s[1]);
    k3b_fastdiv(p.txn, p.fd_m[2], p.fd_s[2]);
    if (SUMS != (p.sums != nullptr) || (SUMS && !FA && p.x_stats != nullptr)) return VS_EINVAL;
    auto kern = k3b_kernel<CK, MT, EPI, SUMS, YT, HS, T, FA, EA>;
    // idempotent one-time opt-in to the full 160 KiB of dynamic LDS (not a stream operation)
    static const hipError_t attr_err =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    // persistent grid: three workgroups per CU, each walking a strided slice of the tile list
    const int per_cu = vs_cfg().k3_wgs_per_cu > 0 ? vs_cfg().k3_wgs_per_cu : 3;   // tuning knob, vs_config.k3_wgs_per_cu (measured: 8->8 @96^3 36.6 / 40.1 / 44.9 / 51.5 us at 3 / 4 / 5 / 6)
    int wg = 256 * per_cu / (row_tiles < per_cu ? row_tiles : per_cu);
    if (wg < 256) wg = 256;
    // one workgroup per tile while the tiles fit; the persistent cap `wg` is a multiple of 8 (kernel: XCD-aware walk).  Never round a
    // small grid down to a multiple of 8: the workgroups that then take two tiles double the latency of the whole launch.
    const int gx = tiles_total < wg ? tiles_total : wg;
    if (EA) {                                            // one tile per workgroup, every workgroup resident while its sample's peers wait for it
        if (gx != tiles_total || (long long)tiles_total * row_tiles > k3b_ea_max_wgs(p.N, p.C, p.M) || !p.ea_sync || !p.ea_fault) return VS_ESHAPE;
        p.ea_items = p.tiles_per_sample * row_tiles;
    }
    hipLaunchKernelGGL(kern, dim3(gx, row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// the fused-apply instantiations: the single-chunk backward-data layers of the 96^3 / 48^3 levels (and their 128^3 / 160^3 counterparts)
template <int CK, int MT, int EPI, bool SUMS, int YT>
constexpr bool k3b_has_fa() {
    // (round 4 also instantiated CK == 32 — the 24^3 / 12^3 levels at one wave per SIMD — which measured slower twice and left the library:
    //  profiles/r04_ab_fused_apply_32ch.json)
    return EPI == EPI_RAW && ((CK == 16 && MT == 16) || (CK == 8 && MT == 16 && !SUMS && YT == 4) || (CK == 16 && MT == 32 && !SUMS && YT == 4));
}

template <typename T, int CK, int MT, int EPI, bool SUMS, int YT = 4>
static int k3b_launch(const G1Params& p, int tiles_total, int row_tiles, hipStream_t stream) {
    if (p.fa_x != nullptr) {
        if constexpr (k3b_has_fa<CK, MT, EPI, SUMS, YT>()) return k3b_launch_t<T, CK, MT, EPI, SUMS, YT, false, true>(p, tiles_total, row_tiles, stream);
        else return VS_ESHAPE;
    }
    if (!SUMS && p.x_stats != nullptr) return k3b_launch_t<T, CK, MT, EPI, SUMS, YT, !SUMS>(p, tiles_total, row_tiles, stream);
    return k3b_launch_t<T, CK, MT, EPI, SUMS, YT, false>(p, tiles_total, row_tiles, stream);
}

// Tall (4x8x16) tiles: measured faster (16->16 @48^3: 19.4 -> 15.4 us) where the layer is one wave of workgroups anyway — fewer,
// fatter workgroups, 2.1x instead of 2.5x halo — and slower (8->8 @96^3: 40 -> 45 us) where workgroups walk many tiles and the
// lower occupancy of the taller tile costs more than its halo saves.  VS_K3_TALL=0/1 forces the choice (tuning / tests).
static inline bool k3b_use_tall(const G1Params& p) {
    const int force = vs_cfg().k3_tall;
    if (force >= 0) return force != 0;
    const long long tall = (long long)((p.D + 3) / 4) * ((p.H + 7) / 8) * p.txn * p.N;
    return tall >= 256 && (long long)p.tiles_per_sample * p.N <= 2048;
}
