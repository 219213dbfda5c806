// Training data pipeline on the device: what the reference runs on 16 CPU workers per step —
// utils/utils.py:220-276 (relabel), 326-383 (CropResize: bounding box, crop + zero pad, skimage resize), 927-968 (MySpatialTransform =
// batchgenerators augment_spatial: rotation / scale / random crop, cubic-spline image and nearest label interpolation), 508-533 / 575-618
// (clip, centre) — as streaming kernels on planar fp32 volumes [D][H][W].  The interpolation arithmetic is scipy.ndimage's (the library both
// skimage.transform.resize and augment_spatial call): gaussian_filter / zoom(grid_mode=True) with mirror boundaries, order-3 spline
// prefilter (mirror initialisation, fp64 like scipy) + map_coordinates with mode 'constant'.  Not tuned: one thread per output voxel / line.
#include "common.h"
#include <limits.h>

__device__ __forceinline__ int dp_mirror(int i, int n) {       // scipy 'mirror': d c b | a b c d | c b a
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i = i < 0 ? -i : i;
    i %= p;
    return i >= n ? p - i : i;
}

// ---- bounding box of label > 0: box = {min z, y, x, max z, y, x} ----------------------------------------------------------------------------
__global__ void dp_bbox_init_kernel(int* box) {
    if (threadIdx.x < 3) box[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) box[threadIdx.x] = -1;
}
__global__ __launch_bounds__(256) void dp_bbox_kernel(const float* __restrict__ lab, int d, int h, int w, int* box) {
    const long long total = (long long)d * h * w;
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {-1, -1, -1};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (lab[i] > 0.f) {
            const int x = (int)(i % w), y = (int)((i / w) % h), z = (int)(i / ((long long)w * h));
            mn[0] = min(mn[0], z); mn[1] = min(mn[1], y); mn[2] = min(mn[2], x);
            mx[0] = max(mx[0], z); mx[1] = max(mx[1], y); mx[2] = max(mx[2], x);
        }
    }
    // one atomic per (workgroup, bound): same-address device atomics retire ~25 ns apart, and one per wave of a 20M-voxel case made this the
    // longest kernel of the pipeline (1.33 ms of 2.66)
    __shared__ int s_mn[4][3], s_mx[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int a = mn[k], b = mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o, 64)); b = max(b, __shfl_xor(b, o, 64)); }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][k] = a; s_mx[threadIdx.x >> 6][k] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const int a = min(min(s_mn[0][k], s_mn[1][k]), min(s_mn[2][k], s_mn[3][k]));
        const int b = max(max(s_mx[0][k], s_mx[1][k]), max(s_mx[2][k], s_mx[3][k]));
        if (a != INT_MAX) atomicMin(box + k, a);
        if (b >= 0) atomicMax(box + 3 + k, b);
    }
}

// ---- relabel: out = target of the first (source -> target) pair whose source equals the label, else 0 ---------------------------------------
struct DpLabelMap { float src[16]; float dst[16]; int n; };
__global__ __launch_bounds__(256) void dp_relabel_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, DpLabelMap m) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float v = in[i];
        float o = 0.f;
        for (int k = 0; k < m.n; ++k) if (v == m.src[k]) o = m.dst[k];          // later pairs win, like the reference's successive assignments
        out[i] = o;
    }
}

// ---- crop + zero pad: dst[z][y][x] = src[z - off + lo] inside [lo, hi) of the source, 0 elsewhere ---------------------------------------------
struct DpCrop { int sd, sh, sw, dd, dh, dw; int lo[3], hi[3], off[3]; };
__global__ __launch_bounds__(256) void dp_crop_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, DpCrop c) {
    const long long total = (long long)c.dd * c.dh * c.dw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % c.dw), y = (int)((i / c.dw) % c.dh), z = (int)(i / ((long long)c.dw * c.dh));
        const int sz = z - c.off[0] + c.lo[0], sy = y - c.off[1] + c.lo[1], sx = x - c.off[2] + c.lo[2];
        const bool in = sz >= c.lo[0] && sz < c.hi[0] && sy >= c.lo[1] && sy < c.hi[1] && sx >= c.lo[2] && sx < c.hi[2];
        dst[i] = in ? src[((long long)sz * c.sh + sy) * c.sw + sx] : 0.f;
    }
}

// ---- scipy.ndimage.gaussian_filter1d(mode='mirror', truncate=4) along one axis -----------------------------------------------------------------
__global__ __launch_bounds__(256) void dp_gauss_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int d, int h, int w, int axis,
                                                           float sigma, int radius) {
    const long long total = (long long)d * h * w;
    const int n = axis == 0 ? d : (axis == 1 ? h : w);
    const long long stride = axis == 0 ? (long long)h * w : (axis == 1 ? w : 1);
    double wsum = 0.0;
    for (int k = -radius; k <= radius; ++k) wsum += exp(-0.5 * (double)k * k / ((double)sigma * sigma));
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pos = (int)((i / stride) % n);
        const long long base = i - (long long)pos * stride;
        double acc = 0.0;
        for (int k = -radius; k <= radius; ++k)
            acc += exp(-0.5 * (double)k * k / ((double)sigma * sigma)) * (double)src[base + (long long)dp_mirror(pos + k, n) * stride];
        dst[i] = (float)(acc / wsum);
    }
}

// ---- scipy.ndimage.zoom(order, mode='mirror', grid_mode=True): input coordinate of output o = (o + 0.5) * n_in / n_out - 0.5 ------------------
// order 0: nearest (floor(x + 0.5)); order 1: trilinear; [clip_lo, clip_hi]: skimage's clip=True (to the input's range), skipped when lo > hi
__global__ __launch_bounds__(256) void dp_zoom_kernel(const float* __restrict__ src, float* __restrict__ dst, int sd, int sh, int sw, int dd, int dh,
                                                     int dw, int order, float clip_lo, float clip_hi) {
    const long long total = (long long)dd * dh * dw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % dw), oy = (int)((i / dw) % dh), oz = (int)(i / ((long long)dw * dh));
        const double cz = ((double)oz + 0.5) * sd / dd - 0.5, cy = ((double)oy + 0.5) * sh / dh - 0.5, cx = ((double)ox + 0.5) * sw / dw - 0.5;
        float v;
        if (order == 0) {
            const int z = dp_mirror((int)floor(cz + 0.5), sd), y = dp_mirror((int)floor(cy + 0.5), sh), x = dp_mirror((int)floor(cx + 0.5), sw);
            v = src[((long long)z * sh + y) * sw + x];
        } else {
            const int z0 = (int)floor(cz), y0 = (int)floor(cy), x0 = (int)floor(cx);
            const double tz = cz - z0, ty = cy - y0, tx = cx - x0;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int z = dp_mirror(z0 + ((k >> 2) & 1), sd), y = dp_mirror(y0 + ((k >> 1) & 1), sh), x = dp_mirror(x0 + (k & 1), sw);
                const double wgt = ((k & 4) ? tz : 1.0 - tz) * ((k & 2) ? ty : 1.0 - ty) * ((k & 1) ? tx : 1.0 - tx);
                acc += wgt * (double)src[((long long)z * sh + y) * sw + x];
            }
            v = (float)acc;
            if (clip_lo <= clip_hi) v = fminf(fmaxf(v, clip_lo), clip_hi);
        }
        dst[i] = v;
    }
}

// ---- min / max of a volume (skimage's clip range): out = {min, max}, caller initialises to {+inf, -inf} ------------------------------------------
__global__ void dp_minmax_init_kernel(float* mm) { if (threadIdx.x == 0) { mm[0] = INFINITY; mm[1] = -INFINITY; } }
__device__ __forceinline__ void dp_atomic_minf(float* a, float v) {          // CAS loop: fp32 atomic min for any sign
    unsigned int* p = (unsigned int*)a; unsigned int old = *p, assumed;
    do { assumed = old; if (__uint_as_float(assumed) <= v) break; old = atomicCAS(p, assumed, __float_as_uint(v)); } while (assumed != old);
}
__device__ __forceinline__ void dp_atomic_maxf(float* a, float v) {
    unsigned int* p = (unsigned int*)a; unsigned int old = *p, assumed;
    do { assumed = old; if (__uint_as_float(assumed) >= v) break; old = atomicCAS(p, assumed, __float_as_uint(v)); } while (assumed != old);
}
__global__ __launch_bounds__(256) void dp_minmax_kernel(const float* __restrict__ x, long long total, float* mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) { lo = fminf(lo, x[i]); hi = fmaxf(hi, x[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    __shared__ float s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {                              // one CAS loop per workgroup and bound
        dp_atomic_minf(mm, fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3])));
        dp_atomic_maxf(mm + 1, fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3])));
    }
}

// ---- order-3 B-spline coefficients (scipy.ndimage.spline_filter1d, mode 'mirror' — what map_coordinates(mode='constant') prefilters with):
// one thread per line along `axis`, fp64 in place.  pole z = sqrt(3) - 2, gain 6, exact mirror initialisation of the causal pass.
__global__ __launch_bounds__(256) void dp_spline3_axis_kernel(double* __restrict__ c, int d, int h, int w, int axis) {
    const int n = axis == 0 ? d : (axis == 1 ? h : w);
    const long long lines = (long long)d * h * w / n;
    const long long stride = axis == 0 ? (long long)h * w : (axis == 1 ? w : 1);
    const long long li = (long long)blockIdx.x * 256 + threadIdx.x;
    if (li >= lines || n < 2) return;
    long long base;
    if (axis == 0) base = li;                                             // (y, x)
    else if (axis == 1) base = (li / w) * (long long)h * w + li % w;      // (z, x)
    else base = li * w;                                                   // (z, y)
    const double z = -0.26794919243112270647;                             // sqrt(3) - 2
    double* a = c + base;
    for (int k = 0; k < n; ++k) a[k * stride] *= 6.0;
    double zn = 1.0;
    for (int k = 0; k < n - 1; ++k) zn *= z;                              // z^(n-1)
    double s = a[0] + zn * a[(long long)(n - 1) * stride];
    double z1 = z, z2 = zn * zn / z;
    for (int k = 1; k < n - 1; ++k) { s += (z1 + z2) * a[k * stride]; z1 *= z; z2 /= z; }
    a[0] = s / (1.0 - zn * zn);
    for (int k = 1; k < n; ++k) a[k * stride] += z * a[(k - 1) * stride];
    a[(long long)(n - 1) * stride] = (z / (z * z - 1.0)) * (z * a[(long long)(n - 2) * stride] + a[(long long)(n - 1) * stride]);
    for (int k = n - 2; k >= 0; --k) a[k * stride] = z * (a[(k + 1) * stride] - a[k * stride]);
}
__global__ __launch_bounds__(256) void dp_widen_kernel(const float* __restrict__ x, double* __restrict__ y, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) y[i] = (double)x[i];
}

// ---- affine resampling: input coordinate of output voxel o = A (o - (P - 1) / 2) + ctr (augment_spatial: A = scale * R^T); scipy
// map_coordinates(mode='constant'): cval where any coordinate leaves [0, n - 1]; order 3 on the spline coefficients (taps mirrored), order 0 nearest
struct DpAffine { double a[9]; double ctr[3]; int sd, sh, sw, pd, ph, pw; };
__device__ __forceinline__ void dp_cubic_w(double t, double (&w)[4]) {
    w[0] = (1 - t) * (1 - t) * (1 - t) / 6.0;
    w[1] = (3 * t * t * t - 6 * t * t + 4) / 6.0;
    w[2] = (-3 * t * t * t + 3 * t * t + 3 * t + 1) / 6.0;
    w[3] = t * t * t / 6.0;
}
template <int ORDER>
__global__ __launch_bounds__(256) void dp_affine_kernel(const void* __restrict__ src, float* __restrict__ dst, DpAffine p, float cval) {
    const long long total = (long long)p.pd * p.ph * p.pw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % p.pw), oy = (int)((i / p.pw) % p.ph), oz = (int)(i / ((long long)p.pw * p.ph));
        const double uz = oz - (p.pd - 1) / 2.0, uy = oy - (p.ph - 1) / 2.0, ux = ox - (p.pw - 1) / 2.0;
        const double cz = p.a[0] * uz + p.a[1] * uy + p.a[2] * ux + p.ctr[0];
        const double cy = p.a[3] * uz + p.a[4] * uy + p.a[5] * ux + p.ctr[1];
        const double cx = p.a[6] * uz + p.a[7] * uy + p.a[8] * ux + p.ctr[2];
        float v = cval;
        if (cz >= 0.0 && cz <= p.sd - 1.0 && cy >= 0.0 && cy <= p.sh - 1.0 && cx >= 0.0 && cx <= p.sw - 1.0) {
            if (ORDER == 0) {
                const int z = (int)floor(cz + 0.5), y = (int)floor(cy + 0.5), x = (int)floor(cx + 0.5);
                v = ((const float*)src)[((long long)z * p.sh + y) * p.sw + x];
            } else {
                const int z0 = (int)floor(cz), y0 = (int)floor(cy), x0 = (int)floor(cx);
                double wz[4], wy[4], wx[4];
                dp_cubic_w(cz - z0, wz); dp_cubic_w(cy - y0, wy); dp_cubic_w(cx - x0, wx);
                const double* co = (const double*)src;
                double acc = 0.0;
                for (int a = 0; a < 4; ++a) {
                    const long long zo = (long long)dp_mirror(z0 - 1 + a, p.sd) * p.sh;
                    for (int b = 0; b < 4; ++b) {
                        const long long yo = (zo + dp_mirror(y0 - 1 + b, p.sh)) * p.sw;
                        double row = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) row += wx[k] * co[yo + dp_mirror(x0 - 1 + k, p.sw)];
                        acc += wz[a] * wy[b] * row;
                    }
                }
                v = (float)acc;
            }
        }
        dst[i] = v;
    }
}

// ---- Clip + CenterIntensities: x = (clamp(x, lo, hi) - subtrahend) / divisor --------------------------------------------------------------------
__global__ __launch_bounds__(256) void dp_clip_center_kernel(float* __restrict__ x, long long total, float lo, float hi, float sub, float div) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) x[i] = (fminf(fmaxf(x[i], lo), hi) - sub) / div;
}

static inline int dp_blocks(long long total) { long long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }
static inline bool dp_dims_ok(int d, int h, int w) { return d > 0 && h > 0 && w > 0 && (double)d * h * w < 2147483648.0; }

extern "C" int vs_data_bbox(const float* label, int d, int h, int w, int* box6, void* stream) {
    if (!label || !box6 || !dp_dims_ok(d, h, w)) return VS_EINVAL;
    hipLaunchKernelGGL(dp_bbox_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, box6);
    const int bb = dp_blocks((long long)d * h * w / 16);
    hipLaunchKernelGGL(dp_bbox_kernel, dim3(bb < 2048 ? bb : 2048), dim3(256), 0, (hipStream_t)stream, label, d, h, w, box6);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_relabel(const float* in, float* out, long long total, const float* sources, const float* targets, int n_pairs, void* stream) {
    if (!in || !out || total <= 0 || n_pairs < 0 || n_pairs > 16 || (n_pairs && (!sources || !targets))) return VS_EINVAL;
    DpLabelMap m{};
    m.n = n_pairs;
    for (int i = 0; i < n_pairs; ++i) { m.src[i] = sources[i]; m.dst[i] = targets[i]; }      // HOST arrays (a handful of label values)
    hipLaunchKernelGGL(dp_relabel_kernel, dim3(dp_blocks(total)), dim3(256), 0, (hipStream_t)stream, in, out, total, m);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_crop_pad(const float* src, float* dst, int sd, int sh, int sw, int dd, int dh, int dw, const int* lo3, const int* hi3,
                                const int* off3, void* stream) {
    if (!src || !dst || !lo3 || !hi3 || !off3 || !dp_dims_ok(sd, sh, sw) || !dp_dims_ok(dd, dh, dw)) return VS_EINVAL;
    DpCrop c{};
    c.sd = sd; c.sh = sh; c.sw = sw; c.dd = dd; c.dh = dh; c.dw = dw;
    const int sdim[3] = {sd, sh, sw};
    for (int k = 0; k < 3; ++k) {
        if (lo3[k] < 0 || hi3[k] > sdim[k] || lo3[k] > hi3[k] || off3[k] < 0) return VS_ESHAPE;
        c.lo[k] = lo3[k]; c.hi[k] = hi3[k]; c.off[k] = off3[k];
    }
    hipLaunchKernelGGL(dp_crop_pad_kernel, dim3(dp_blocks((long long)dd * dh * dw)), dim3(256), 0, (hipStream_t)stream, src, dst, c);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_gaussian_axis(const float* src, float* dst, int d, int h, int w, int axis, float sigma, void* stream) {
    if (!src || !dst || src == dst || !dp_dims_ok(d, h, w) || axis < 0 || axis > 2 || !(sigma > 0.f)) return VS_EINVAL;
    const int radius = (int)(4.0f * sigma + 0.5f);
    hipLaunchKernelGGL(dp_gauss_axis_kernel, dim3(dp_blocks((long long)d * h * w)), dim3(256), 0, (hipStream_t)stream, src, dst, d, h, w, axis, sigma, radius);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_minmax(const float* x, long long total, float* minmax2, void* stream) {
    if (!x || !minmax2 || total <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(dp_minmax_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, minmax2);
    const int mb = dp_blocks(total / 16);
    hipLaunchKernelGGL(dp_minmax_kernel, dim3(mb < 2048 ? mb : 2048), dim3(256), 0, (hipStream_t)stream, x, total, minmax2);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_zoom(const float* src, float* dst, int sd, int sh, int sw, int dd, int dh, int dw, int order, float clip_lo, float clip_hi,
                            void* stream) {
    if (!src || !dst || !dp_dims_ok(sd, sh, sw) || !dp_dims_ok(dd, dh, dw) || (order != 0 && order != 1)) return VS_EINVAL;
    hipLaunchKernelGGL(dp_zoom_kernel, dim3(dp_blocks((long long)dd * dh * dw)), dim3(256), 0, (hipStream_t)stream, src, dst, sd, sh, sw, dd, dh, dw, order,
                       clip_lo, clip_hi);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_spline3_prefilter(const float* x, double* coef, int d, int h, int w, void* stream) {
    if (!x || !coef || !dp_dims_ok(d, h, w)) return VS_EINVAL;
    const long long total = (long long)d * h * w;
    hipLaunchKernelGGL(dp_widen_kernel, dim3(dp_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, coef, total);
    const int dims[3] = {d, h, w};
    for (int axis = 0; axis < 3; ++axis) {
        if (dims[axis] < 2) continue;
        hipLaunchKernelGGL(dp_spline3_axis_kernel, dim3(dp_blocks(total / dims[axis])), dim3(256), 0, (hipStream_t)stream, coef, d, h, w, axis);
    }
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_affine_sample(const void* src, float* dst, int sd, int sh, int sw, int pd, int ph, int pw, const double* a9, const double* ctr3,
                                     int order, float cval, void* stream) {
    if (!src || !dst || !a9 || !ctr3 || !dp_dims_ok(sd, sh, sw) || !dp_dims_ok(pd, ph, pw) || (order != 0 && order != 3)) return VS_EINVAL;
    DpAffine p{};
    for (int i = 0; i < 9; ++i) p.a[i] = a9[i];            // HOST arrays
    for (int i = 0; i < 3; ++i) p.ctr[i] = ctr3[i];
    p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw;
    const int blocks = dp_blocks((long long)pd * ph * pw);
    if (order == 0) hipLaunchKernelGGL(dp_affine_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, p, cval);
    else hipLaunchKernelGGL(dp_affine_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, p, cval);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_data_clip_center(float* x, long long total, float lo, float hi, float subtrahend, float divisor, void* stream) {
    if (!x || total <= 0 || !(lo <= hi) || divisor == 0.f) return VS_EINVAL;
    hipLaunchKernelGGL(dp_clip_center_kernel, dim3(dp_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, total, lo, hi, subtrahend, divisor);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
