// Kernels of the reference's `*_GS` model family (joint_model.py:17-33 GSNorm3d, :54-99 DoubleConv_GS / Up_GS / Down_GS / Conv_GS,
// :307-346 Segmentation_GS) that the InstanceNorm U-Net does not need: the channel-group normalisation, trilinear upsampling by an
// integer factor, and a standalone two-class softmax.  Nothing in the reference instantiates these classes, so the kernels are plain
// streaming code (one thread per voxel and channel fragment), not tuned: channels-last tensors, 16-byte fragments.
#include "common.h"

// ---- GSNorm3d: y[c] = x[c] / (sum over c's group of x + 1e-4) -----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gsnorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int c, int interval) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (row = (n, voxel), group)
    const int groups = c / interval;
    if (i >= rows * groups) return;
    const long long row = i / groups;
    const int gidx = (int)(i - row * groups);
    const T* px = x + row * c + gidx * interval;
    T* py = y + row * c + gidx * interval;
    float s = 1e-4f;
    for (int j = 0; j < interval; ++j) s += ET<T>::ld(px + j);
    const float inv = 1.f / s;
    for (int j = 0; j < interval; ++j) ET<T>::st(py + j, ET<T>::ld(px + j) * inv);
}

// dx[c] = g[c] / S - (sum_j g[j] x[j]) / S^2,  S = sum_j x[j] + 1e-4
template <typename T>
__global__ __launch_bounds__(256) void gsnorm_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x, T* __restrict__ dx, long long rows,
                                                        int c, int interval) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int groups = c / interval;
    if (i >= rows * groups) return;
    const long long row = i / groups;
    const int gidx = (int)(i - row * groups);
    const size_t o = (size_t)row * c + gidx * interval;
    float s = 1e-4f, gx = 0.f;
    for (int j = 0; j < interval; ++j) { const float xv = ET<T>::ld(x + o + j); s += xv; gx += ET<T>::ld(g + o + j) * xv; }
    const float inv = 1.f / s, k = gx * inv * inv;
    for (int j = 0; j < interval; ++j) ET<T>::st(dx + o + j, ET<T>::ld(g + o + j) * inv - k);
}

// ---- torch.nn.Upsample(scale_factor = s, mode = 'trilinear') [align_corners = False] ---------------------------------------------
// source coordinate of output index o: max(0, (o + 0.5) / s - 0.5); neighbours i0 = floor, i1 = min(i0 + 1, size - 1)
__device__ __forceinline__ void up_src(int o, int s, int size, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) / (float)s - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + 1 < size ? i0 + 1 : size - 1;
    l1 = src - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int d, int h, int w, int c, int s) {
    constexpr int EPL = ET<T>::EPL;
    const int frags = c / EPL;
    const long long total = (long long)n * d * s * h * s * w * s * frags;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % frags);
    long long v = i / frags;
    const int ox = (int)(v % (w * s)); v /= w * s;
    const int oy = (int)(v % (h * s)); v /= h * s;
    const int oz = (int)(v % (d * s));
    const int b = (int)(v / (d * s));
    int z0, z1, y0, y1, x0, x1;
    float lz, ly, lx;
    up_src(oz, s, d, z0, z1, lz); up_src(oy, s, h, y0, y1, ly); up_src(ox, s, w, x0, x1, lx);
    float acc[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zz = (k & 4) ? z1 : z0, yy = (k & 2) ? y1 : y0, xx = (k & 1) ? x1 : x0;
        const float wgt = ((k & 4) ? lz : 1.f - lz) * ((k & 2) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
        float fv[EPL];
        frag_unpack(*(const u32x4*)(x + ((((size_t)b * d + zz) * h + yy) * w + xx) * c + f * EPL), fv, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[j] += wgt * fv[j];
    }
    *(u32x4*)(y + ((((size_t)b * d * s + oz) * h * s + oy) * w * s + ox) * c + f * EPL) = frag_pack(acc, (T*)nullptr);
}

// backward: every output voxel adds its gradient, weighted, to its 8 source voxels (fp32 atomics into `acc`, zeroed by the caller's launch);
// a second pass rounds to the storage type
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ g, float* __restrict__ acc, int n, int d, int h, int w, int c, int s) {
    constexpr int EPL = ET<T>::EPL;
    const int frags = c / EPL;
    const long long total = (long long)n * d * s * h * s * w * s * frags;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % frags);
    long long v = i / frags;
    const int ox = (int)(v % (w * s)); v /= w * s;
    const int oy = (int)(v % (h * s)); v /= h * s;
    const int oz = (int)(v % (d * s));
    const int b = (int)(v / (d * s));
    int z0, z1, y0, y1, x0, x1;
    float lz, ly, lx;
    up_src(oz, s, d, z0, z1, lz); up_src(oy, s, h, y0, y1, ly); up_src(ox, s, w, x0, x1, lx);
    float gv[EPL];
    frag_unpack(*(const u32x4*)(g + ((((size_t)b * d * s + oz) * h * s + oy) * w * s + ox) * c + f * EPL), gv, (T*)nullptr);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zz = (k & 4) ? z1 : z0, yy = (k & 2) ? y1 : y0, xx = (k & 1) ? x1 : x0;
        const float wgt = ((k & 4) ? lz : 1.f - lz) * ((k & 2) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
        if (wgt == 0.f) continue;
        float* dst = acc + ((((size_t)b * d + zz) * h + yy) * w + xx) * c + f * EPL;
#pragma unroll
        for (int j = 0; j < EPL; ++j) atomicAdd(dst + j, wgt * gv[j]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void round_store_kernel(const float* __restrict__ acc, T* __restrict__ out, long long frags_total) {
    constexpr int EPL = ET<T>::EPL;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= frags_total) return;
    float v[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) v[j] = acc[i * EPL + j];
    *(u32x4*)(out + i * EPL) = frag_pack(v, (T*)nullptr);
}

// ---- nn.Softmax(dim=1) over two classes: channels-last logits (channels 0, 1 of c) -> planar fp32 probabilities [n][2][voxels] ------
template <typename T>
__global__ __launch_bounds__(256) void softmax2_fwd_kernel(const T* __restrict__ x, float* __restrict__ prob, int n, long long voxels, int c) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * voxels) return;
    const long long b = i / voxels, v = i - b * voxels;
    const float l0 = ET<T>::ld(x + i * c), l1 = ET<T>::ld(x + i * c + 1);
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float inv = 1.f / (e0 + e1);
    prob[(b * 2 + 0) * voxels + v] = e0 * inv;
    prob[(b * 2 + 1) * voxels + v] = e1 * inv;
}

static int gs_check(const void* a, const void* b, long long rows, int c, int dtype) {
    if (!a || !b || rows <= 0) return VS_EINVAL;
    if (c <= 0 || c % 8) return VS_ESHAPE;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    return VS_OK;
}

extern "C" int vs_gsnorm_fwd(const void* x, void* y, long long rows, int c, int num_group, int dtype, void* stream) {
    int rc = gs_check(x, y, rows, c, dtype);
    if (rc) return rc;
    if (num_group <= 0 || c % num_group) return VS_ESHAPE;
    const long long total = rows * num_group;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(gsnorm_fwd_kernel<T>, dim3(vs_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, rows, c, c / num_group);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_gsnorm_bwd(const void* g, const void* x, void* dx, long long rows, int c, int num_group, int dtype, void* stream) {
    int rc = gs_check(x, dx, rows, c, dtype);
    if (rc) return rc;
    if (!g || num_group <= 0 || c % num_group) return VS_ESHAPE;
    const long long total = rows * num_group;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(gsnorm_bwd_kernel<T>, dim3(vs_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)g, (const T*)x, (T*)dx, rows, c, c / num_group);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_upsample_trilinear_fwd(const void* x, void* y, int n, int d, int h, int w, int c, int scale, int dtype, void* stream) {
    int rc = gs_check(x, y, (long long)n * d * h * w, c, dtype);
    if (rc) return rc;
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0 || scale < 1 || scale > 16) return VS_ESHAPE;
    if ((double)n * d * h * w * scale * scale * scale * c >= 2147483648.0 * 4) return VS_ESHAPE;
    const int epl = dtype == VS_F32 ? 4 : 8;
    const long long total = (long long)n * d * scale * h * scale * w * scale * (c / epl);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(upsample_fwd_kernel<T>, dim3(vs_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n, d, h, w, c, scale);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_upsample_trilinear_bwd(const void* g, void* dx, float* scratch, int n, int d, int h, int w, int c, int scale, int dtype, void* stream) {
    int rc = gs_check(g, dx, (long long)n * d * h * w, c, dtype);
    if (rc) return rc;
    if (!scratch || ((uintptr_t)scratch & 15)) return VS_EINVAL;
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0 || scale < 1 || scale > 16) return VS_ESHAPE;
    const int epl = dtype == VS_F32 ? 4 : 8;
    const long long in_elems = (long long)n * d * h * w * c;
    rc = vs_zero_fill(scratch, ((in_elems * 4 + 15) / 16) * 16, stream);
    if (rc) return rc;
    const long long total = (long long)n * d * scale * h * scale * w * scale * (c / epl);
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(vs_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)g, scratch, n, d, h, w, c, scale);
    });
    VS_CHECK_LAUNCH();
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(round_store_kernel<T>, dim3(vs_ceil_div(in_elems / epl, 256)), dim3(256), 0, (hipStream_t)stream, scratch, (T*)dx, in_elems / epl);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_softmax2_fwd(const void* logits, float* prob, int n, long long voxels, int c, int dtype, void* stream) {
    int rc = gs_check(logits, prob, (long long)n * voxels, c, dtype);
    if (rc) return rc;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(softmax2_fwd_kernel<T>, dim3(vs_ceil_div((long long)n * voxels, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)logits, prob, n, voxels, c);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}
