// InstanceNorm3d(affine=False)+ReLU helpers on channels-last tensors: standalone statistics, materialisation
// (with the optional U-Net skip add), and the two-pass backward (per-(n,c) reductions, then the apply pass).
// All are HBM-bound streaming kernels: 16-byte fragments, one workgroup = 256 threads laid out as
// (C/EPL channel fragments) x (voxel sub-rows), fp64 atomics only once per (block, channel, statistic).
#include "common.h"

#define NB_MAX_C 256

template <typename T>
struct RowIter {
    // thread -> (channel fragment fx, voxel sub-row fy); a block covers `rows_per_it` voxels per iteration
    int fx, fy, frags, rows_per_it;
    __device__ RowIter(int c) {
        frags = c / ET<T>::EPL;
        fx = threadIdx.x % frags;
        fy = threadIdx.x / frags;
        rows_per_it = 256 / frags;
    }
    __device__ bool active() const { return fy < rows_per_it; }
};

__device__ __forceinline__ void load_mean_rstd(const double* stats, int n, int c, double inv_count, float eps,
                                               float* s_mean, float* s_rstd) {
    const size_t pairs = (size_t)gridDim.y * c;          // every kernel of this file runs blockIdx.y = sample over all N samples
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        float m = 0.f, r = 1.f;
        if (stats) stats_to_mean_rstd(stats, (size_t)n * c + i, pairs, inv_count, eps, m, r);
        s_mean[i] = m;
        s_rstd[i] = r;
    }
}

// block-level reduction of per-thread channel partials (EPL channels x NS statistics) + fp64 atomics
template <typename T, int NS>
__device__ __forceinline__ void reduce_and_atomic(const RowIter<T>& it, double (&part)[ET<T>::EPL][NS], double* out,
                                                  int n, int c, double* s_red /* [256][EPL*NS] */) {
    constexpr int EPL = ET<T>::EPL;
#pragma unroll
    for (int j = 0; j < EPL; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) s_red[threadIdx.x * (EPL * NS) + j * NS + s] = it.active() ? part[j][s] : 0.0;
    __syncthreads();
    // thread t < c*NS sums one (channel, statistic) over the voxel sub-rows
    for (int o = threadIdx.x; o < c * NS; o += 256) {
        const int ch = o / NS, s = o - ch * NS;
        const int fx = ch / EPL, j = ch - fx * EPL;
        double tot = 0.0;
        for (int fy = 0; fy < it.rows_per_it; ++fy) tot += s_red[(fy * it.frags + fx) * (EPL * NS) + j * NS + s];
        static_assert(NS == 2, "statistics come in pairs");
        stat_add(out, (size_t)n * c + ch, (size_t)gridDim.y * c, s, tot);
    }
}

// ---- statistics -------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stats_kernel(const T* __restrict__ x, double* __restrict__ stats, long long voxels, int c) {
    constexpr int EPL = ET<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_red = (double*)smem;
    RowIter<T> it(c);
    const int n = blockIdx.y;
    double part[EPL][2];
#pragma unroll
    for (int j = 0; j < EPL; ++j) part[j][0] = part[j][1] = 0.0;
    if (it.active()) {
        const T* base = x + (size_t)n * voxels * c + it.fx * EPL;
        for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < voxels; v += (long long)gridDim.x * it.rows_per_it) {
            float f[EPL];
            frag_unpack(*(const u32x4*)(base + v * c), f, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) { part[j][0] += f[j]; part[j][1] += (double)f[j] * f[j]; }
        }
    }
    reduce_and_atomic<T, 2>(it, part, stats, n, c, s_red);
}

// ---- materialise relu(norm(x)) [+ relu(norm(x2))] ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void in_relu_fwd_kernel(const T* __restrict__ x, const double* xs, const T* __restrict__ x2,
                                                          const double* x2s, T* __restrict__ out, long long voxels, int c,
                                                          double inv_count, float eps) {
    constexpr int EPL = ET<T>::EPL;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C], s_m2[NB_MAX_C], s_r2[NB_MAX_C];
    const int n = blockIdx.y;
    load_mean_rstd(xs, n, c, inv_count, eps, s_m, s_r);
    if (x2) load_mean_rstd(x2s, n, c, inv_count, eps, s_m2, s_r2);
    __syncthreads();
    RowIter<T> it(c);
    if (!it.active()) return;
    const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
    for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < voxels; v += (long long)gridDim.x * it.rows_per_it) {
        float f[EPL], o[EPL];
        frag_unpack(*(const u32x4*)(x + sample + v * c), f, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            float t = f[j];
            if (xs) { t = (t - s_m[it.fx * EPL + j]) * s_r[it.fx * EPL + j]; t = t > 0.f ? t : 0.f; }
            o[j] = t;
        }
        if (x2) {
            frag_unpack(*(const u32x4*)(x2 + sample + v * c), f, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                float t = f[j];
                if (x2s) { t = (t - s_m2[it.fx * EPL + j]) * s_r2[it.fx * EPL + j]; t = t > 0.f ? t : 0.f; }
                o[j] += t;
            }
        }
        *(u32x4*)(out + sample + v * c) = frag_pack(o, (T*)nullptr);
    }
}

// ---- backward: reductions ------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void in_relu_bwd_reduce_body(const T* __restrict__ g, const T* __restrict__ x, const double* xs,
                                                        double* __restrict__ sums, long long voxels, int c,
                                                        double inv_count, float eps) {
    constexpr int EPL = ET<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_red = (double*)smem;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C];
    const int n = blockIdx.y;
    RowIter<T> it(c);
    const bool act = it.active();
    const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
    const long long stride = (long long)gridDim.x * it.rows_per_it;
    long long v = (long long)blockIdx.x * it.rows_per_it + it.fy;
    // as in the apply kernel below: UN (g, x) fragment pairs in flight per thread, the first batch requested before the tables
    constexpr int UN = 4;
    u32x4 gq[UN], xq[UN];
    auto request = [&](long long v0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long vv = v0 + u * stride;
            const size_t e = (act && vv < voxels) ? sample + vv * c : sample - it.fx * EPL;
            gq[u] = *(const u32x4*)(g + e);
            xq[u] = *(const u32x4*)(x + e);
        }
    };
    request(v);
    load_mean_rstd(xs, n, c, inv_count, eps, s_m, s_r);
    __syncthreads();
    double part[EPL][2];
#pragma unroll
    for (int j = 0; j < EPL; ++j) part[j][0] = part[j][1] = 0.0;
    if (act) {
        float m[EPL], r[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { m[j] = s_m[it.fx * EPL + j]; r[j] = s_r[it.fx * EPL + j]; }
        for (; v < voxels; v += UN * stride) {
            u32x4 gc[UN], xc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { gc[u] = gq[u]; xc[u] = xq[u]; }
            request(v + UN * stride);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (v + u * stride >= voxels) break;
                float fg[EPL], fx_[EPL];
                frag_unpack(gc[u], fg, (T*)nullptr);
                frag_unpack(xc[u], fx_, (T*)nullptr);
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const float xh = (fx_[j] - m[j]) * r[j];
                    const float gm = xh > 0.f ? fg[j] : 0.f;
                    part[j][0] += gm;
                    part[j][1] += (double)gm * xh;
                }
            }
        }
    }
    reduce_and_atomic<T, 2>(it, part, sums, n, c, s_red);
}
template <typename T>
__global__ __launch_bounds__(256) void in_relu_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ x, const double* xs,
                                                                 double* __restrict__ sums, long long voxels, int c,
                                                                 double inv_count, float eps) {
    in_relu_bwd_reduce_body<T>(g, x, xs, sums, voxels, c, inv_count, eps);
}
// The additive U-Net skip (joint_model.py:380,382) sends ONE gradient g to two lazy operands: blockIdx.z picks the operand, so the
// reduce and the apply of both are one launch each instead of two.
struct InBwdPair { const void* x[2]; const double* xs[2]; double* sums[2]; void* gx[2]; };
template <typename T>
__global__ __launch_bounds__(256) void in_relu_bwd_reduce2_kernel(const T* __restrict__ g, const InBwdPair a, long long voxels, int c,
                                                                  double inv_count, float eps) {
    const int op = blockIdx.z;
    in_relu_bwd_reduce_body<T>(g, (const T*)a.x[op], a.xs[op], a.sums[op], voxels, c, inv_count, eps);
}

// ---- backward: apply -----------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void in_relu_bwd_apply_body(const T* __restrict__ g, const T* __restrict__ x, const double* xs,
                                                       const double* __restrict__ sums, T* __restrict__ gx,
                                                       long long voxels, int c, double inv_count, float eps, const T* __restrict__ add = nullptr) {
    // Streaming pass, 3 tensors: a thread keeps UN (g, x) fragment pairs in flight and requests the next batch before it
    // works on the current one; the first batch is requested before the statistics tables are built, so the kernel's
    // start-up is one memory round trip, not two (most of its 58 launches per step are small and start-up bound).
    constexpr int EPL = ET<T>::EPL;
    constexpr int UN = 4;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C], s_a[NB_MAX_C], s_b[NB_MAX_C];
    const int n = blockIdx.y;
    RowIter<T> it(c);
    const bool act = it.active();
    const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
    const long long stride = (long long)gridDim.x * it.rows_per_it;
    long long v = (long long)blockIdx.x * it.rows_per_it + it.fy;
    u32x4 gq[UN], xq[UN], aq[UN];
    auto request = [&](long long v0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long vv = v0 + u * stride;
            const size_t e = (act && vv < voxels) ? sample + vv * c : sample - it.fx * EPL;     // clamped lanes re-read voxel 0
            gq[u] = *(const u32x4*)(g + e);
            xq[u] = *(const u32x4*)(x + e);
            if (add != nullptr) aq[u] = *(const u32x4*)(add + e);                               // uniform branch: a second gradient of x to sum in
        }
    };
    request(v);
    load_mean_rstd(xs, n, c, inv_count, eps, s_m, s_r);
    for (int i = threadIdx.x; i < c; i += 256) {
        double sv[2];
        stat_load(sums, (size_t)n * c + i, (size_t)gridDim.y * c, sv);
        s_a[i] = (float)(sv[0] * inv_count);
        s_b[i] = (float)(sv[1] * inv_count);
    }
    __syncthreads();
    if (!act) return;
    float m[EPL], r[EPL], a[EPL], b[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int ch = it.fx * EPL + j;
        m[j] = s_m[ch]; r[j] = s_r[ch]; a[j] = s_a[ch]; b[j] = s_b[ch];
    }
    for (; v < voxels; v += UN * stride) {
        u32x4 gc[UN], xc[UN], ac[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { gc[u] = gq[u]; xc[u] = xq[u]; if (add != nullptr) ac[u] = aq[u]; }
        request(v + UN * stride);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long vv = v + u * stride;
            if (vv >= voxels) break;
            float fg[EPL], fx_[EPL], o[EPL];
            frag_unpack(gc[u], fg, (T*)nullptr);
            frag_unpack(xc[u], fx_, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) o[j] = vs_in_bwd_apply1(fg[j], fx_[j], m[j], r[j], a[j], b[j]);
            if (add != nullptr) {
                float fa[EPL];
                frag_unpack(ac[u], fa, (T*)nullptr);
#pragma unroll
                for (int j = 0; j < EPL; ++j) o[j] = ET<T>::rnd(o[j]) + fa[j];      // what autograd's add of the two stored gradients gives
            }
            *(u32x4*)(gx + sample + vv * c) = frag_pack(o, (T*)nullptr);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void in_relu_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ x, const double* xs,
                                                                const double* __restrict__ sums, T* __restrict__ gx,
                                                                long long voxels, int c, double inv_count, float eps, const T* __restrict__ add) {
    in_relu_bwd_apply_body<T>(g, x, xs, sums, gx, voxels, c, inv_count, eps, add);
}
template <typename T>
__global__ __launch_bounds__(256) void in_relu_bwd_apply2_kernel(const T* __restrict__ g, const InBwdPair a, long long voxels, int c,
                                                                 double inv_count, float eps) {
    const int op = blockIdx.z;
    in_relu_bwd_apply_body<T>(g, (const T*)a.x[op], a.xs[op], a.sums[op], (T*)a.gx[op], voxels, c, inv_count, eps);
}

// ---- bias gradient ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ g, float* __restrict__ db, long long rows, int c, int c_real) {
    constexpr int EPL = ET<T>::EPL;
    __shared__ double s_red[256 * 8];
    RowIter<T> it(c);
    double part[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) part[j] = 0.0;
    if (it.active()) {
        for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < rows; v += (long long)gridDim.x * it.rows_per_it) {
            float f[EPL];
            frag_unpack(*(const u32x4*)(g + v * c + it.fx * EPL), f, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) part[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) s_red[threadIdx.x * EPL + j] = it.active() ? part[j] : 0.0;
    __syncthreads();
    for (int ch = threadIdx.x; ch < c_real; ch += 256) {
        const int fx = ch / EPL, j = ch - fx * EPL;
        double tot = 0.0;
        for (int fy = 0; fy < it.rows_per_it; ++fy) tot += s_red[(fy * it.frags + fx) * EPL + j];
        atomicAdd(db + ch, (float)tot);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
static int check_cl(const void* a, int n, long long voxels, int c, int dtype) {
    if (!a || n <= 0 || voxels <= 0) return VS_EINVAL;
    if (c <= 0 || c % 8 || c > NB_MAX_C || (256 % (c / 8)) ) return VS_ESHAPE;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    return VS_OK;
}
static int row_blocks(long long voxels, int c, int dtype, bool atomics = true) {
    const int frags = c / (dtype == VS_F32 ? 4 : 8);
    const int rpi = 256 / frags;
    long long b = (voxels + rpi - 1) / rpi;
    // each block should stream >= ~16 iterations; cap at 8 blocks per CU overall.  Small tensors (the 24^3 / 48^3 skip merges: 28-108 blocks
    // of 16 dependent rounds = 21 us for 2.6 MB) are latency-bound instead: 4 rounds per block, four times the blocks.
    // (kernels that end in fp64 atomics keep 16: more blocks cost more same-address atomics than the shorter loop saves)
    const int rounds = (!atomics && (double)voxels * c <= 2097152.0) ? 4 : 16;
    long long cap = (voxels + (long long)rpi * rounds - 1) / ((long long)rpi * rounds);
    if (cap < 1) cap = 1;
    if (cap > 2048) cap = 2048;
    return (int)(b < cap ? b : cap);
}
static size_t red_lds(int dtype, int ns) { return (size_t)256 * (dtype == VS_F32 ? 4 : 8) * ns * sizeof(double); }


extern "C" int vs_instnorm_stats(const void* x, double* stats, int n, long long voxels, int c, int dtype, void* stream) {
    int rc = check_cl(x, n, voxels, c, dtype);
    if (rc) return rc;
    if (!stats) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype), n);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(stats_kernel<T>, grid, dim3(256), red_lds(dtype, 2), (hipStream_t)stream, (const T*)x, stats, voxels, c);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_instnorm_relu_fwd(const void* x, const double* x_stats, const void* x2, const double* x2_stats,
                                    void* out, int n, long long voxels, int c, int dtype, float eps, void* stream) {
    int rc = check_cl(x, n, voxels, c, dtype);
    if (rc) return rc;
    if (!out) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype, false), n);
    const double inv = 1.0 / (double)voxels;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(in_relu_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, x_stats, (const T*)x2, x2_stats, (T*)out, voxels, c, inv, eps);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_instnorm_relu_bwd_reduce(const void* g, const void* x, const double* x_stats, double* sums, int n,
                                           long long voxels, int c, int dtype, float eps, void* stream) {
    int rc = check_cl(x, n, voxels, c, dtype);
    if (rc) return rc;
    if (!g || !x_stats || !sums) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype), n);
    const double inv = 1.0 / (double)voxels;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(in_relu_bwd_reduce_kernel<T>, grid, dim3(256), red_lds(dtype, 2), (hipStream_t)stream, (const T*)g, (const T*)x, x_stats, sums, voxels, c, inv, eps);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_instnorm_relu_bwd_apply(const void* g, const void* x, const double* x_stats, const double* sums,
                                          void* gx, int n, long long voxels, int c, int dtype, float eps, void* stream) {
    return vs_instnorm_relu_bwd_apply_add(g, x, x_stats, sums, nullptr, gx, n, voxels, c, dtype, eps, stream);
}

extern "C" int vs_instnorm_relu_bwd_apply_add(const void* g, const void* x, const double* x_stats, const double* sums, const void* add,
                                              void* gx, int n, long long voxels, int c, int dtype, float eps, void* stream) {
    int rc = check_cl(x, n, voxels, c, dtype);
    if (rc) return rc;
    if (!g || !x_stats || !sums || !gx) return VS_EINVAL;
    // one batch of 4 rows per thread where the volume allows it, at most ~8 workgroups per CU over the n samples
    const int rpi = 256 / (c / (dtype == VS_F32 ? 4 : 8));
    long long gb = (voxels + (long long)rpi * 4 - 1) / ((long long)rpi * 4);
    const long long gcap = 2048 / n > 0 ? 2048 / n : 1;
    if (gb > gcap) gb = gcap;
    dim3 grid((unsigned)gb, n);
    const double inv = 1.0 / (double)voxels;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(in_relu_bwd_apply_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g, (const T*)x, x_stats, sums, (T*)gx, voxels, c, inv, eps, (const T*)add);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_instnorm_relu_bwd_pair(const void* g, const void* x1, const double* x1_stats, double* sums1, void* gx1,
                                         const void* x2, const double* x2_stats, double* sums2, void* gx2, int n, long long voxels,
                                         int c, int dtype, float eps, void* stream) {
    int rc = check_cl(x1, n, voxels, c, dtype);
    if (rc) return rc;
    if (!g || !x2 || !x1_stats || !x2_stats || !sums1 || !sums2 || !gx1 || !gx2) return VS_EINVAL;
    InBwdPair a;
    a.x[0] = x1; a.x[1] = x2; a.xs[0] = x1_stats; a.xs[1] = x2_stats; a.sums[0] = sums1; a.sums[1] = sums2; a.gx[0] = gx1; a.gx[1] = gx2;
    const double inv = 1.0 / (double)voxels;
    dim3 grid(row_blocks(voxels, c, dtype), n, 2);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(in_relu_bwd_reduce2_kernel<T>, grid, dim3(256), red_lds(dtype, 2), (hipStream_t)stream, (const T*)g, a, voxels, c, inv, eps);
    });
    VS_CHECK_LAUNCH();
    const int rpi = 256 / (c / (dtype == VS_F32 ? 4 : 8));
    long long gb = (voxels + (long long)rpi * 4 - 1) / ((long long)rpi * 4);
    const long long gcap = 2048 / (2 * n) > 0 ? 2048 / (2 * n) : 1;
    if (gb > gcap) gb = gcap;
    dim3 grid2((unsigned)gb, n, 2);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(in_relu_bwd_apply2_kernel<T>, grid2, dim3(256), 0, (hipStream_t)stream, (const T*)g, a, voxels, c, inv, eps);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_bias_grad(const void* g, float* db, long long rows, int c_ch, int c_real, int dtype, void* stream) {
    return vs_bias_grad_acc(g, db, rows, c_ch, c_real, dtype, 0, stream);
}

extern "C" int vs_bias_grad_acc(const void* g, float* db, long long rows, int c_ch, int c_real, int dtype, int accumulate, void* stream) {
    int rc = check_cl(g, 1, rows, c_ch, dtype);
    if (rc) return rc;
    if (!db || c_real <= 0 || c_real > c_ch) return VS_EINVAL;
    if (!accumulate) {
        hipError_t e = vs_zero_async(db, sizeof(float) * c_real, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(VS_DET_BUILD ? 1 : row_blocks(rows, c_ch, dtype));     // deterministic mode: one block, one fp32 atomic per channel
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(bias_grad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g, db, rows, c_ch, c_real);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// General normalisation + activation (vaeseg.h: vs_norm_*): BatchNorm3d (joint_model.py:13, norm_type=2 — the class default, which
// no entry point passes) with its affine pair and running statistics, InstanceNorm3d, ReLU or Softplus (joint_model.py:38, soft=True).
// No entry point of the reference reaches these settings, so they are NOT fused into the conv kernels: the conv writes its raw
// output and (sum, sumsq) statistics as always, one tiny kernel turns the statistics into per-(n,c) mean / rstd tables (pooled over
// the batch for BatchNorm, running statistics updated in the same kernel), and three streaming passes do forward, backward
// reduction and backward apply.  The activation that leaves is a stored tensor (the next conv takes it as a non-lazy input).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float na_act(float u, int act) {
    if (act == VS_ACT_RELU) return u > 0.f ? u : 0.f;
    return u > 20.f ? u : log1pf(expf(u));                 // torch.nn.Softplus(beta=1, threshold=20)
}
__device__ __forceinline__ float na_dact(float u, int act) {
    if (act == VS_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    return u > 20.f ? 1.f : 1.f / (1.f + expf(-u));
}

// one workgroup; tables float[n*c] each.  BatchNorm training: batch mean / biased variance normalise, the running pair takes the
// unbiased variance (torch.nn.BatchNorm3d), num_batches_tracked is bumped.
__global__ __launch_bounds__(256) void norm_tables_kernel(const double* __restrict__ stats, int n, int c, double count, int mode, float eps,
                                                          float momentum, float* running_mean, float* running_var, long long* tracked,
                                                          int c_real, float* __restrict__ mean, float* __restrict__ rstd) {
    const size_t pairs = (size_t)n * c;
    if (mode == VS_NORM_NONE) {
        for (int i = threadIdx.x; i < n * c; i += 256) { mean[i] = 0.f; rstd[i] = 1.f; }
        return;
    }
    if (mode == VS_NORM_INSTANCE) {
        for (int i = threadIdx.x; i < n * c; i += 256) {
            double st[2];
            stat_load(stats, (size_t)i, pairs, st);
            float m, r;
            pair_to_mean_rstd(st, 1.0 / count, eps, m, r);
            mean[i] = m; rstd[i] = r;
        }
        return;
    }
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float m = 0.f, r = 1.f;
        if (mode == VS_NORM_BATCH) {
            double s = 0.0, q = 0.0;
            for (int i = 0; i < n; ++i) {
                double st[2];
                stat_load(stats, (size_t)i * c + ch, pairs, st);
                s += st[0]; q += st[1];
            }
            const double tot = count * n;
            const double md = s / tot;
            double var = q / tot - md * md;
            if (var < 0.0) var = 0.0;
            m = (float)md;
            r = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean != nullptr && ch < c_real) {
                running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * m;
                running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)(tot > 1.0 ? var * tot / (tot - 1.0) : var);
            }
        } else if (ch < c_real) {                               // VS_NORM_BATCH_EVAL
            m = running_mean[ch];
            r = (float)(1.0 / sqrt((double)running_var[ch] + (double)eps));
        }
        for (int i = 0; i < n; ++i) { mean[i * c + ch] = m; rstd[i * c + ch] = r; }
    }
    if (mode == VS_NORM_BATCH && tracked != nullptr && threadIdx.x == 0) *tracked += 1;
}

// scale / shift of u = xhat * gamma + beta written as x * sc + sh; channels beyond c_real are padding and stay zero
__device__ __forceinline__ void na_load_tables(const float* mean, const float* rstd, const float* gamma, const float* beta, int n, int c,
                                               int c_real, float* s_m, float* s_r, float* s_g, float* s_b) {
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        s_m[i] = mean[n * c + i];
        s_r[i] = rstd[n * c + i];
        s_g[i] = i < c_real ? (gamma != nullptr ? gamma[i] : 1.f) : 0.f;
        s_b[i] = (i < c_real && beta != nullptr) ? beta[i] : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const T* __restrict__ x, const float* mean, const float* rstd, const float* gamma,
                                                           const float* beta, T* __restrict__ y, long long voxels, int c, int c_real, int act) {
    constexpr int EPL = ET<T>::EPL;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C], s_g[NB_MAX_C], s_b[NB_MAX_C];
    const int n = blockIdx.y;
    na_load_tables(mean, rstd, gamma, beta, n, c, c_real, s_m, s_r, s_g, s_b);
    __syncthreads();
    RowIter<T> it(c);
    if (!it.active()) return;
    float m[EPL], r[EPL], ga[EPL], be[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) { const int ch = it.fx * EPL + j; m[j] = s_m[ch]; r[j] = s_r[ch]; ga[j] = s_g[ch]; be[j] = s_b[ch]; }
    const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
    for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < voxels; v += (long long)gridDim.x * it.rows_per_it) {
        float f[EPL], o[EPL];
        frag_unpack(*(const u32x4*)(x + sample + v * c), f, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const float u = (f[j] - m[j]) * r[j] * ga[j] + be[j];
            o[j] = it.fx * EPL + j < c_real ? na_act(u, act) : 0.f;
        }
        *(u32x4*)(y + sample + v * c) = frag_pack(o, (T*)nullptr);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_act_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ x, const float* mean,
                                                                  const float* rstd, const float* gamma, const float* beta,
                                                                  double* __restrict__ sums, long long voxels, int c, int c_real, int act) {
    constexpr int EPL = ET<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_red = (double*)smem;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C], s_g[NB_MAX_C], s_b[NB_MAX_C];
    const int n = blockIdx.y;
    na_load_tables(mean, rstd, gamma, beta, n, c, c_real, s_m, s_r, s_g, s_b);
    __syncthreads();
    RowIter<T> it(c);
    double part[EPL][2];
#pragma unroll
    for (int j = 0; j < EPL; ++j) part[j][0] = part[j][1] = 0.0;
    if (it.active()) {
        float m[EPL], r[EPL], ga[EPL], be[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { const int ch = it.fx * EPL + j; m[j] = s_m[ch]; r[j] = s_r[ch]; ga[j] = s_g[ch]; be[j] = s_b[ch]; }
        const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
        for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < voxels; v += (long long)gridDim.x * it.rows_per_it) {
            float fg[EPL], fx_[EPL];
            frag_unpack(*(const u32x4*)(g + sample + v * c), fg, (T*)nullptr);
            frag_unpack(*(const u32x4*)(x + sample + v * c), fx_, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const float xh = (fx_[j] - m[j]) * r[j];
                const float da = fg[j] * na_dact(xh * ga[j] + be[j], act);
                part[j][0] += da;
                part[j][1] += (double)da * xh;
            }
        }
    }
    reduce_and_atomic<T, 2>(it, part, sums, n, c, s_red);
}

// one workgroup: coef[n*c][3] = (k, m1, m2) of dx = k * (da - m1 - xhat * m2); dgamma / dbeta (fp32 [c_real], nullable)
__global__ __launch_bounds__(256) void norm_act_bwd_finish_kernel(const double* __restrict__ sums, int n, int c, int c_real, double count, int mode,
                                                                  const float* rstd, const float* gamma, float* __restrict__ coef,
                                                                  float* dgamma, float* dbeta) {
    const size_t pairs = (size_t)n * c;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double st[2];
            stat_load(sums, (size_t)i * c + ch, pairs, st);
            s1 += st[0]; s2 += st[1];
            if (mode == VS_NORM_INSTANCE) {
                coef[((size_t)i * c + ch) * 3 + 1] = (float)(st[0] / count);
                coef[((size_t)i * c + ch) * 3 + 2] = (float)(st[1] / count);
            }
        }
        const float ga = ch < c_real ? (gamma != nullptr ? gamma[ch] : 1.f) : 0.f;
        for (int i = 0; i < n; ++i) {
            float* co = coef + ((size_t)i * c + ch) * 3;
            co[0] = ga * rstd[i * c + ch];
            if (mode == VS_NORM_BATCH) { co[1] = (float)(s1 / (count * n)); co[2] = (float)(s2 / (count * n)); }
            else if (mode != VS_NORM_INSTANCE) { co[1] = 0.f; co[2] = 0.f; }        // running statistics / no normalisation: no batch terms
        }
        if (ch < c_real) {
            if (dgamma != nullptr) dgamma[ch] = (float)s2;
            if (dbeta != nullptr) dbeta[ch] = (float)s1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_act_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ x, const float* mean,
                                                                 const float* rstd, const float* gamma, const float* beta,
                                                                 const float* __restrict__ coef, T* __restrict__ dx, long long voxels, int c,
                                                                 int c_real, int act) {
    constexpr int EPL = ET<T>::EPL;
    __shared__ float s_m[NB_MAX_C], s_r[NB_MAX_C], s_g[NB_MAX_C], s_b[NB_MAX_C], s_k[NB_MAX_C], s_1[NB_MAX_C], s_2[NB_MAX_C];
    const int n = blockIdx.y;
    na_load_tables(mean, rstd, gamma, beta, n, c, c_real, s_m, s_r, s_g, s_b);
    for (int i = threadIdx.x; i < c; i += 256) {
        const float* co = coef + ((size_t)n * c + i) * 3;
        s_k[i] = co[0]; s_1[i] = co[1]; s_2[i] = co[2];
    }
    __syncthreads();
    RowIter<T> it(c);
    if (!it.active()) return;
    float m[EPL], r[EPL], ga[EPL], be[EPL], k[EPL], m1[EPL], m2[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int ch = it.fx * EPL + j;
        m[j] = s_m[ch]; r[j] = s_r[ch]; ga[j] = s_g[ch]; be[j] = s_b[ch]; k[j] = s_k[ch]; m1[j] = s_1[ch]; m2[j] = s_2[ch];
    }
    const size_t sample = (size_t)n * voxels * c + it.fx * EPL;
    for (long long v = (long long)blockIdx.x * it.rows_per_it + it.fy; v < voxels; v += (long long)gridDim.x * it.rows_per_it) {
        float fg[EPL], fx_[EPL], o[EPL];
        frag_unpack(*(const u32x4*)(g + sample + v * c), fg, (T*)nullptr);
        frag_unpack(*(const u32x4*)(x + sample + v * c), fx_, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const float xh = (fx_[j] - m[j]) * r[j];
            const float da = fg[j] * na_dact(xh * ga[j] + be[j], act);
            o[j] = k[j] * (da - m1[j] - xh * m2[j]);
        }
        *(u32x4*)(dx + sample + v * c) = frag_pack(o, (T*)nullptr);
    }
}

static int na_check(const void* x, const float* mean, const float* rstd, int n, long long voxels, int c, int c_real, int act, int dtype) {
    int rc = check_cl(x, n, voxels, c, dtype);
    if (rc) return rc;
    if (!mean || !rstd || c_real <= 0 || c_real > c) return VS_EINVAL;
    if (act != VS_ACT_RELU && act != VS_ACT_SOFTPLUS) return VS_EINVAL;
    return VS_OK;
}

extern "C" int vs_norm_tables(const double* stats, int n, int c, int c_real, double count, int mode, float eps, float momentum,
                              float* running_mean, float* running_var, long long* num_batches_tracked, float* mean, float* rstd, void* stream) {
    if (!mean || !rstd || n <= 0 || c <= 0 || c_real <= 0 || c_real > c || count <= 0) return VS_EINVAL;
    if (mode != VS_NORM_INSTANCE && mode != VS_NORM_BATCH && mode != VS_NORM_BATCH_EVAL && mode != VS_NORM_NONE) return VS_EINVAL;
    if ((mode == VS_NORM_INSTANCE || mode == VS_NORM_BATCH) && !stats) return VS_EINVAL;
    if (mode == VS_NORM_BATCH_EVAL && (!running_mean || !running_var)) return VS_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return VS_EINVAL;
    hipLaunchKernelGGL(norm_tables_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, n, c, count, mode, eps, momentum, running_mean,
                       running_var, num_batches_tracked, c_real, mean, rstd);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_norm_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, int n,
                               long long voxels, int c, int c_real, int act, int dtype, void* stream) {
    int rc = na_check(x, mean, rstd, n, voxels, c, c_real, act, dtype);
    if (rc) return rc;
    if (!y) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype, false), n);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(norm_act_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, mean, rstd, gamma, beta, (T*)y, voxels, c, c_real, act);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_norm_act_bwd_reduce(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      double* sums, int n, long long voxels, int c, int c_real, int act, int dtype, void* stream) {
    int rc = na_check(x, mean, rstd, n, voxels, c, c_real, act, dtype);
    if (rc) return rc;
    if (!g || !sums) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype), n);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(norm_act_bwd_reduce_kernel<T>, grid, dim3(256), red_lds(dtype, 2), (hipStream_t)stream, (const T*)g, (const T*)x, mean, rstd,
                           gamma, beta, sums, voxels, c, c_real, act);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_norm_act_bwd_finish(const double* sums, int n, int c, int c_real, double count, int mode, const float* rstd, const float* gamma,
                                      float* coef, float* dgamma, float* dbeta, void* stream) {
    if (!sums || !rstd || !coef || n <= 0 || c <= 0 || c_real <= 0 || c_real > c || count <= 0) return VS_EINVAL;
    if (mode != VS_NORM_INSTANCE && mode != VS_NORM_BATCH && mode != VS_NORM_BATCH_EVAL && mode != VS_NORM_NONE) return VS_EINVAL;
    hipLaunchKernelGGL(norm_act_bwd_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, n, c, c_real, count, mode, rstd, gamma, coef, dgamma, dbeta);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_norm_act_bwd_apply(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     const float* coef, void* dx, int n, long long voxels, int c, int c_real, int act, int dtype, void* stream) {
    int rc = na_check(x, mean, rstd, n, voxels, c, c_real, act, dtype);
    if (rc) return rc;
    if (!g || !coef || !dx) return VS_EINVAL;
    dim3 grid(row_blocks(voxels, c, dtype, false), n);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(norm_act_bwd_apply_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g, (const T*)x, mean, rstd, gamma, beta,
                           coef, (T*)dx, voxels, c, c_real, act);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}
