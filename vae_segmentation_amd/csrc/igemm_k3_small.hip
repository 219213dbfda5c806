// 3x3x3 conv (forward / backward-data) for the deepest U-Net levels: tiny volumes (<= 224 voxels per sample, e.g. 3^3,
// 4^3, 6^3), many channels (multiple of 32), bf16.
//
// On these layers the general tile kernel wastes its 4x4x16 voxel tile (3 of 16 x-columns used at 3^3) and is bound by
// staging arithmetic and by the latency of streaming weight fragments (PMC: 9.4k VALU vs 0.9k MFMA instructions per wave).
// Here the GEMM columns are the flattened voxels of one sample (ceil(V/16) column groups, no padding waste in x), the
// whole zero-padded sample volume of a 32-channel chunk sits in LDS, and the four waves of a workgroup split K: each wave
// takes a different channel chunk per round, so the workgroup runs four independent weight streams (each wave requests all
// 27 fragments of its chunk up front) and nothing is loaded twice.  The four partial accumulators are summed through LDS
// in a fixed order (bitwise reproducible), then the usual epilogue (store, optional bias, fp64 statistics or the fused
// InstanceNorm-backward sums) runs.
#include "igemm.h"
#include "igemm_dispatch.h"

#define KS_MAXCG 14

#ifdef VS_STAMPS   // diagnostic build only: per-phase s_memtime stamps of wave 0 of every workgroup (never in the product)
__device__ unsigned long long g_ks_stamps[256 * 16];
#define STAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_ks_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_; } } while (0)
extern "C" int vs_debug_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ks_stamps), sizeof(unsigned long long) * n);
}
#else
#define STAMP(i)
#endif
#define KS_LDS_RED 0                  // float[4][64][2]  (stats flush)
#define KS_LDS_MEAN 2048              // float[256] mean, float[256] rstd of the input; then the same for the mask tensor
#define KS_LDS_SLOTS (2048 + 4 * 1024)

__global__ __launch_bounds__(256) void k3_small_kernel(const G1Params p, int PD_, int PH_, int PW_, int ncg) {
    typedef unsigned short T;
    constexpr int CK = 32, CKB = 64, NKG = 27;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_red = (float*)(smem + KS_LDS_RED);
    float* s_mean = (float*)(smem + KS_LDS_MEAN);
    float* s_rstd = s_mean + 256;
    float* s_mkm = s_rstd + 256;
    float* s_mkr = s_mkm + 256;
    char* s_slots = smem + KS_LDS_SLOTS;
    const int PV = PD_ * PH_ * PW_;                  // padded voxels
    const int slot_bytes = PV * CKB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int n = blockIdx.x, rb = blockIdx.y;
    STAMP(0);
    const int V = p.D * p.H * p.W;
    const bool has_stats = p.x_stats != nullptr;
    const T* __restrict__ xin = (const T*)p.x;

    // ---- per-thread staging table: up to 8 fragments (16 B = 8 channels of one padded voxel) per chunk ----
    constexpr int SB = 8;
    int g_off[SB], l_off[SB];
    bool real[SB], inlist[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const int u = tid + b * 256;
        const int pv = u >> 2, part = u & 3;
        const int px = pv % PW_, py = (pv / PW_) % PH_, pz = pv / (PW_ * PH_);
        inlist[b] = u < PV * 4;
        real[b] = inlist[b] && px >= 1 && px <= p.W && py >= 1 && py <= p.H && pz >= 1 && pz <= p.D;
        g_off[b] = real[b] ? ((((n * p.D + pz - 1) * p.H + py - 1) * p.W + px - 1) * p.C + part * 8) : 0;
        l_off[b] = pv * CKB + part * 16;
    }
    // ---- per-lane column geometry ----
    int cbase[KS_MAXCG];
#pragma unroll
    for (int cg = 0; cg < KS_MAXCG; ++cg) {
        int v = cg * 16 + col;
        if (v >= V) v = 0;                               // padded column: reads voxel 0, never stored
        const int x = v % p.W, y = (v / p.W) % p.H, z = v / (p.W * p.H);
        cbase[cg] = ((z * PH_ + y) * PW_ + x) * CKB + g * 16;
    }

    f32x4 acc[KS_MAXCG];
#pragma unroll
    for (int cg = 0; cg < KS_MAXCG; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};

    const u32x4* __restrict__ wp = (const u32x4*)p.wp;
    STAMP(1);

    for (int round = 0; round * 4 < p.nch; ++round) {
        // ---- stage up to four channel chunks (one per wave's slot), activation applied, halo zero ----
        {
            // all global loads of the round (up to 4 chunks x 8 fragments) are in flight before the first is consumed;
            // in round 0 they are issued BEFORE the statistics tables are computed, so that latency is paid once
            u32x4 vals[4][SB];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ch = round * 4 + s < p.nch ? round * 4 + s : p.nch - 1;
#pragma unroll
                for (int b = 0; b < SB; ++b) vals[s][b] = *(const u32x4*)(xin + g_off[b] + ch * CK);
            }
            STAMP(2);
            if (round == 0) {
            for (int c = tid; c < p.C; c += 256) {
                float m = 0.f, r = 1.f;
                if (has_stats) stats_to_mean_rstd(p.x_stats + ((size_t)n * p.C + c) * 2, p.inv_count_in, p.eps, m, r);
                s_mean[c] = m; s_rstd[c] = r;
            }
            if (p.sums != nullptr) {
                for (int c = tid; c < p.M; c += 256) {
                    float m, r;
                    stats_to_mean_rstd(p.mask_stats + ((size_t)n * p.M + c) * 2, p.inv_count_out, p.eps, m, r);
                    s_mkm[c] = m; s_mkr[c] = r;
                }
            }
            }
            STAMP(3);
            __syncthreads();                             // tables visible / previous round's slots fully consumed
            STAMP(4);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ch = round * 4 + s;
                if (ch < p.nch) {
#pragma unroll
                    for (int b = 0; b < SB; ++b) {
                        if (inlist[b]) {
                            u32x4 val = u32x4{0u, 0u, 0u, 0u};
                            if (real[b]) {
                                val = vals[s][b];
                                if (has_stats) val = act_transform<T, CK>(val, s_mean, s_rstd, ch * CK + (tid & 3) * 8);
                            }
                            *(u32x4*)(s_slots + s * slot_bytes + l_off[b]) = val;
                        }
                    }
                }
            }
        }
        // ---- each wave multiplies its own chunk: 27 k-groups; all weight fragments are requested before the barrier so
        //      their latency overlaps the other waves' staging writes ----
        STAMP(5);
        const int ch = round * 4 + wave;
        const int chc = ch < p.nch ? ch : p.nch - 1;
        u32x4 a[NKG];
        {
            const u32x4* wch = wp + ((size_t)(rb * p.nch + chc) * NKG) * 64 + lane;
#pragma unroll
            for (int kg = 0; kg < NKG; ++kg) a[kg] = wch[kg * 64];
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
        if (ch < p.nch) {
            const char* slot = s_slots + wave * slot_bytes;
#pragma unroll
            for (int kg = 0; kg < NKG; ++kg) {
                const int dz = kg / 9, dy = (kg / 3) % 3, dx = kg % 3;
                const int toff = ((dz * PH_ + dy) * PW_ + dx) * CKB;
                // all B fragments of the k-group are requested before the first MFMA (one LDS latency per k-group instead
                // of one per MFMA: with a single wave per SIMD nothing else hides it)
                u32x4 b[KS_MAXCG];
#pragma unroll
                for (int cg = 0; cg < KS_MAXCG; ++cg)
                    if (cg < ncg) b[cg] = *(const u32x4*)(slot + cbase[cg] + toff);
#pragma unroll
                for (int cg = 0; cg < KS_MAXCG; ++cg)
                    if (cg < ncg) acc[cg] = mfma16(a[kg], b[cg], acc[cg], (T*)nullptr);
            }
        }
    }

    // ---- sum the four waves' partial accumulators (fixed order) ----
    STAMP(8);
    __syncthreads();
    STAMP(9);
    f32x4* s_part = (f32x4*)s_slots;                     // [wave][cg][lane]
#pragma unroll
    for (int cg = 0; cg < KS_MAXCG; ++cg)
        if (cg < ncg) s_part[(wave * KS_MAXCG + cg) * 64 + lane] = acc[cg];
    __syncthreads();

    STAMP(10);
    // ---- epilogue: wave w finishes column groups w, w+4, ... ----
    T* __restrict__ yout = (T*)p.y;
    const int row = rb * 16 + 4 * g;
    const bool rvalid = row < p.M;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && rvalid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias[row + r];
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int cg = wave; cg < ncg; cg += 4) {
        f32x4 tot = s_part[(0 * KS_MAXCG + cg) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 o = s_part[(w * KS_MAXCG + cg) * 64 + lane];
            tot[0] += o[0]; tot[1] += o[1]; tot[2] += o[2]; tot[3] += o[3];
        }
        const int v = cg * 16 + col;
        if (!(rvalid && v < V)) continue;
        float vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] = round_bf(tot[r] + bv[r]);
        const size_t e = ((size_t)n * V + v) * p.M + row;
        u32x2 pk;
        pk[0] = (unsigned int)f2bf(vv[0]) | ((unsigned int)f2bf(vv[1]) << 16);
        pk[1] = (unsigned int)f2bf(vv[2]) | ((unsigned int)f2bf(vv[3]) << 16);
        *(u32x2*)(yout + e) = pk;
        if (p.sums != nullptr) {
            const u32x2 xx = *(const u32x2*)((const T*)p.mask_x + e);
            float xv[4];
            xv[0] = __uint_as_float(xx[0] << 16); xv[1] = __uint_as_float(xx[0] & 0xffff0000u);
            xv[2] = __uint_as_float(xx[1] << 16); xv[3] = __uint_as_float(xx[1] & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (xv[r] - s_mkm[row + r]) * s_mkr[row + r];
                const float gm = xh > 0.f ? vv[r] : 0.f;
                ssum[r] += gm; ssq[r] += gm * xh;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[r] += vv[r]; ssq[r] += vv[r] * vv[r]; }
        }
    }
    STAMP(11);
    double* const red_dst = p.sums != nullptr ? p.sums : p.y_stats;
    if (red_dst != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = ssum[r], q = ssq[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            if (col == 0) {
                s_red[(wave * 64 + 4 * g + r) * 2 + 0] = s;
                s_red[(wave * 64 + 4 * g + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int lr = tid >> 1, st = tid & 1;
            const int rr = rb * 16 + lr;
            if (rr < p.M) {
                const double tot = (double)s_red[(0 * 64 + lr) * 2 + st] + (double)s_red[(1 * 64 + lr) * 2 + st] +
                                   (double)s_red[(2 * 64 + lr) * 2 + st] + (double)s_red[(3 * 64 + lr) * 2 + st];
                atomicAdd(red_dst + ((size_t)n * p.M + rr) * 2 + st, tot);
            }
        }
    }
    STAMP(12);
}

// returns VS_OK if launched, K3_SMALL_NA if the shape is not one this kernel handles (caller falls back to the tile kernel)
int k3_small_try(const G1Params& p, int dtype, hipStream_t stream) {
    if (dtype != VS_BF16 || p.C % 32 || p.C > 256 || p.M > 256) return K3_SMALL_NA;
    const int V = p.D * p.H * p.W;
    const int pd = p.D + 2, ph = p.H + 2, pw = p.W + 2;
    const int PV = pd * ph * pw;
    if (V > KS_MAXCG * 16 || PV * 4 > 8 * 256) return K3_SMALL_NA;
    const size_t slot = (size_t)PV * 64;
    size_t lds = KS_LDS_SLOTS + 4 * slot;
    const size_t part = (size_t)4 * KS_MAXCG * 64 * 16;
    if (KS_LDS_SLOTS + part > lds) lds = KS_LDS_SLOTS + part;
    if (lds > 160 * 1024) return K3_SMALL_NA;
    static const hipError_t attr_err =
        hipFuncSetAttribute((const void*)k3_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    const int ncg = (V + 15) / 16;
    hipLaunchKernelGGL(k3_small_kernel, dim3(p.N, p.rb_total), dim3(256), lds, stream, p, pd, ph, pw, ncg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VS_OK : (int)e;
}
