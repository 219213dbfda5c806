// Small / streaming kernels of the training step: softmax backward, label prep, VAE bottleneck (fc, reparam, KL),
// Dice / BCE losses, multi-tensor SGD / Adam / EMA.
#include "common.h"

// ---- deterministic build (common.h): a property of the library, queried by the package ----------------------------------------------
extern "C" int vs_get_deterministic(void) { return VS_DET_BUILD; }

extern "C" int vs_version(void) { return VS_VERSION; }
extern "C" int vs_stat_slots(void) { return VS_STAT_SLOTS; }
extern "C" int vs_stat_interleaved(void) { return VS_STAT_INTERLEAVE; }
extern "C" const char* vs_strerror(int code) {
    switch (code) {
        case VS_OK: return "ok";
        case VS_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case VS_ESHAPE: return "unsupported shape (channels must be 8, 16 or a multiple of 32 up to 256; even dims for stride 2; batch <= 16 for wgrad)";
        case VS_EDTYPE: return "unsupported dtype (VS_F32, VS_BF16 or VS_F16)";
        case VS_EWORKSPACE: return "workspace too small";
        case VS_EALIGN: return "pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown libvaeseg error";
    }
}

#define GRID1D(n) dim3((unsigned)((((n) + 255) / 256) < 65536LL * 16 ? (((n) + 255) / 256) : 65536LL * 16))

// ---- softmax (2 classes) backward ----------------------------------------------------------------
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ out, long long frags, float p, unsigned long long seed) {
    constexpr int EPL = ET<T>::EPL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < frags; i += (long long)gridDim.x * blockDim.x) {
        float f[EPL];
        frag_unpack(*(const u32x4*)(x + i * EPL), f, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) f[j] *= dropout_scale(seed, (unsigned long long)i * EPL + j, p);
        *(u32x4*)(out + i * EPL) = frag_pack(f, (T*)nullptr);
    }
}
extern "C" int vs_dropout(const void* x, void* out, long long count, float p, unsigned long long seed, int dtype, void* stream) {
    if (!x || !out || count <= 0 || p < 0.f || p >= 1.f) return VS_EINVAL;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    const int epl = dtype == VS_F32 ? 4 : 8;
    if (count % epl) return VS_ESHAPE;
    const long long frags = count / epl;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(dropout_kernel<T>, GRID1D(frags), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, frags, p, seed);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, long long count, float p, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        mask[i] = dropout_scale(seed, (unsigned long long)i, p);
}
extern "C" int vs_dropout_mask(float* mask, long long count, float p, unsigned long long seed, void* stream) {
    if (!mask || count <= 0 || p < 0.f || p >= 1.f) return VS_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, GRID1D(count), dim3(256), 0, (hipStream_t)stream, mask, count, p, seed);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

template <typename T>
__global__ void softmax2_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ gprob, const T* __restrict__ gcl, T* __restrict__ gl,
                                    long long voxels, int c_pad, long long total, float drop_p, unsigned long long drop_seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / voxels, v = i - n * voxels;
        const float p0 = prob[(n * 2 + 0) * voxels + v], p1 = prob[(n * 2 + 1) * voxels + v];
        float g0 = 0.f, g1 = 0.f;
        if (gprob) { g0 = gprob[(n * 2 + 0) * voxels + v]; g1 = gprob[(n * 2 + 1) * voxels + v]; }
        if (gcl) {                  // + the gradient that arrived through the channels-last copy of the probabilities
            float f2[ET<T>::EPL];
            frag_unpack(*(const u32x4*)(gcl + i * c_pad), f2, (T*)nullptr);
            g0 += f2[0]; g1 += f2[1];
        }
        const float dot = p0 * g0 + p1 * g1;
        float f[ET<T>::EPL];
#pragma unroll
        for (int j = 0; j < ET<T>::EPL; ++j) f[j] = 0.f;
        f[0] = p0 * (g0 - dot);
        f[1] = p1 * (g1 - dot);
        if (drop_p > 0.f) {          // backward of the logit dropout fused into the out_block epilogue
            f[0] *= dropout_scale(drop_seed, ((unsigned long long)n * 2 + 0) * voxels + v, drop_p);
            f[1] *= dropout_scale(drop_seed, ((unsigned long long)n * 2 + 1) * voxels + v, drop_p);
        }
        T* o = gl + i * c_pad;
        *(u32x4*)o = frag_pack(f, (T*)nullptr);
        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
        for (int c0 = ET<T>::EPL; c0 < c_pad; c0 += ET<T>::EPL) *(u32x4*)(o + c0) = z;
    }
}

extern "C" int vs_softmax2_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad,
                               int dtype, void* stream) {
    return vs_softmax2_dropout_bwd(prob, gprob, glogit, n, voxels, c_pad, dtype, 0.f, 0ull, stream);
}

extern "C" int vs_softmax2_dropout_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad,
                                       int dtype, float drop_p, unsigned long long drop_seed, void* stream) {
    if (!gprob) return VS_EINVAL;
    return vs_softmax2_cl_bwd(prob, gprob, nullptr, glogit, n, voxels, c_pad, dtype, drop_p, drop_seed, stream);
}

extern "C" int vs_softmax2_cl_bwd(const float* prob, const float* gprob, const void* gprob_cl, void* glogit, int n, long long voxels,
                                  int c_pad, int dtype, float drop_p, unsigned long long drop_seed, void* stream) {
    if (!prob || (!gprob && !gprob_cl) || !glogit || n <= 0 || voxels <= 0 || c_pad % 8 || c_pad <= 0) return VS_EINVAL;
    const long long total = (long long)n * voxels;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(softmax2_bwd_kernel<T>, GRID1D(total), dim3(256), 0, (hipStream_t)stream, prob, gprob, (const T*)gprob_cl, (T*)glogit, voxels, c_pad, total, drop_p, drop_seed);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- softmax over n_class = 1..8 classes as its own pass ------------------------------------------
// The two-class case (every configuration BASELINE names) has the fused out_block epilogue; a label set with more structures
// (main_source.py:92-93: n_class = 1 + number of --pan_index entries) takes the plain 3x3x3 conv and these two kernels.
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&f)[8]) {
    constexpr int EPL = ET<T>::EPL;
#pragma unroll
    for (int k = 0; k < 8 / EPL; ++k) {
        float part[EPL];
        frag_unpack(*(const u32x4*)(p + k * EPL), part, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) f[k * EPL + j] = part[j];
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&f)[8]) {
    constexpr int EPL = ET<T>::EPL;
#pragma unroll
    for (int k = 0; k < 8 / EPL; ++k) {
        float part[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) part[j] = f[k * EPL + j];
        *(u32x4*)(p + k * EPL) = frag_pack(part, (T*)nullptr);
    }
}

template <typename T>
__global__ void softmaxn_fwd_kernel(const T* __restrict__ logits, float* __restrict__ prob, long long voxels, int c_pad, int nc, long long total,
                                    float drop_p, unsigned long long drop_seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / voxels, v = i - n * voxels;
        float l[8];
        load8(logits + i * c_pad, l);
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < nc) {
                if (drop_p > 0.f) l[c] *= dropout_scale(drop_seed, ((unsigned long long)n * nc + c) * voxels + v, drop_p);
                mx = fmaxf(mx, l[c]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            l[c] = c < nc ? expf(l[c] - mx) : 0.f;
            sum += l[c];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < nc) prob[(n * nc + c) * voxels + v] = l[c] * inv;
    }
}

template <typename T>
__global__ void softmaxn_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ gprob, T* __restrict__ gl, long long voxels, int c_pad, int nc,
                                    long long total, float drop_p, unsigned long long drop_seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / voxels, v = i - n * voxels;
        float p[8], g[8], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            p[c] = c < nc ? prob[(n * nc + c) * voxels + v] : 0.f;
            g[c] = c < nc ? gprob[(n * nc + c) * voxels + v] : 0.f;
            dot += p[c] * g[c];
        }
        float f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f[c] = p[c] * (g[c] - dot);
            if (drop_p > 0.f && c < nc) f[c] *= dropout_scale(drop_seed, ((unsigned long long)n * nc + c) * voxels + v, drop_p);
        }
        T* o = gl + i * c_pad;
        store8(o, f);
        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
        for (int c0 = 8; c0 < c_pad; c0 += ET<T>::EPL) *(u32x4*)(o + c0) = z;
    }
}

static int softmaxn_check(const void* a, const void* b, int n, long long voxels, int c_pad, int n_class, int dtype, float drop_p) {
    if (!a || !b || n <= 0 || voxels <= 0 || drop_p < 0.f || drop_p >= 1.f) return VS_EINVAL;
    if (c_pad <= 0 || c_pad % 8 || n_class < 1 || n_class > 8) return VS_ESHAPE;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    return VS_OK;
}

extern "C" int vs_softmax_cl_fwd(const void* logits, float* prob, int n, long long voxels, int c_pad, int n_class, int dtype, float drop_p,
                                 unsigned long long drop_seed, void* stream) {
    const int rc = softmaxn_check(logits, prob, n, voxels, c_pad, n_class, dtype, drop_p);
    if (rc) return rc;
    const long long total = (long long)n * voxels;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(softmaxn_fwd_kernel<T>, GRID1D(total), dim3(256), 0, (hipStream_t)stream, (const T*)logits, prob, voxels, c_pad, n_class, total, drop_p, drop_seed);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_softmax_cl_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad, int n_class, int dtype,
                                 float drop_p, unsigned long long drop_seed, void* stream) {
    const int rc = softmaxn_check(prob, glogit, n, voxels, c_pad, n_class, dtype, drop_p);
    if (rc) return rc;
    if (!gprob) return VS_EINVAL;
    const long long total = (long long)n * voxels;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(softmaxn_bwd_kernel<T>, GRID1D(total), dim3(256), 0, (hipStream_t)stream, prob, gprob, (T*)glogit, voxels, c_pad, n_class, total, drop_p, drop_seed);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- one-hot / binarize ---------------------------------------------------------------------------
__global__ void onehot_kernel(const float* __restrict__ label, float* __restrict__ out, long long voxels, int n_class, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / voxels, v = i - n * voxels;
        const int k = (int)label[i];          // .long() truncation, as label.type(LongTensor)
        for (int c = 0; c < n_class; ++c) out[(n * n_class + c) * voxels + v] = c == k ? 1.f : 0.f;
    }
}
extern "C" int vs_onehot(const float* label, float* out, int n, long long voxels, int n_class, void* stream) {
    if (!label || !out || n <= 0 || voxels <= 0 || n_class <= 0) return VS_EINVAL;
    const long long total = (long long)n * voxels;
    hipLaunchKernelGGL(onehot_kernel, GRID1D(total), dim3(256), 0, (hipStream_t)stream, label, out, voxels, n_class, total);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// argmax over the channel axis of a planar (n, c, voxels) tensor -> its one-hot, the hard masks of the validation Dice (utils/evaluation.py:58-64:
// torch.argmax -> scatter_).  Ties go to the FIRST maximal channel, as torch.argmax resolves them; a NaN channel wins (as torch's max does).
__global__ void hard_onehot_kernel(const float* __restrict__ x, float* __restrict__ out, long long voxels, int c, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / voxels, v = i - b * voxels;
        const float* xp = x + (size_t)b * c * voxels + v;
        float best = xp[0];
        int arg = 0;
        for (int k = 1; k < c; ++k) {
            const float val = xp[(size_t)k * voxels];
            if (val > best || (val != val && best == best)) { best = val; arg = k; }
        }
        float* op = out + (size_t)b * c * voxels + v;
        for (int k = 0; k < c; ++k) op[(size_t)k * voxels] = k == arg ? 1.f : 0.f;
    }
}
extern "C" int vs_hard_onehot(const float* x, float* out, int n, int n_class, long long voxels, void* stream) {
    if (!x || !out || n <= 0 || voxels <= 0 || n_class <= 0) return VS_EINVAL;
    const long long total = (long long)n * voxels;
    hipLaunchKernelGGL(hard_onehot_kernel, GRID1D(total), dim3(256), 0, (hipStream_t)stream, x, out, voxels, n_class, total);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ void binarize_kernel(const float* __restrict__ a, float* __restrict__ out, long long count, int mode, float lo, float hi) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const float v = a[i];
        float r;
        if (mode == 0) r = v >= 0.5f ? 1.f : 0.f;
        else r = v > hi ? 1.f : (v < lo ? 0.f : v);
        out[i] = r;
    }
}
extern "C" int vs_binarize(const float* a, float* out, long long count, int mode, float lo, float hi, void* stream) {
    if (!a || !out || count <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(binarize_kernel, GRID1D(count), dim3(256), 0, (hipStream_t)stream, a, out, count, mode, lo, hi);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- fully connected --------------------------------------------------------------------------------
__device__ __forceinline__ long long phys_index(int k, int pc, int pv) { return pc > 0 ? (long long)(k % pv) * pc + k / pv : k; }
__device__ __forceinline__ float ld_any(const void* p, int dtype, long long i) {
    if (dtype == VS_F32) return ((const float*)p)[i];
    if (dtype == VS_BF16) return bf2f(((const unsigned short*)p)[i]);
    return (float)((const vs_half*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int dtype, long long i, float v) {
    if (dtype == VS_F32) ((float*)p)[i] = v;
    else if (dtype == VS_BF16) ((unsigned short*)p)[i] = f2bf(v);
    else ((vs_half*)p)[i] = (vs_half)v;
}
#define LIN_MAXB 16

// one workgroup per output j: y[b][j] = act(bias[j] + sum_k W[j][k] x[b][phys(k)]).  NB = batch rounded up to 1/2/4/8/16 at compile
// time (padding rows re-read row batch-1 and are dropped), k unrolled by 4: 4 weight + 4*NB activation loads are in flight per
// thread instead of one dependent load per FMA (the bottleneck GEMV is latency-, not bandwidth-bound: 27 k-steps per thread).
// Round 6: (1) 16-byte weight loads, four consecutive k per lane and load — the bottleneck widths finish in ONE round of <= 8 loads per thread (dword loads in
// two dependent rounds streamed the frozen fc2's 14 MB at 1.1 TB/s: 12.6 us); (2) a PERMUTED activation (pc > 0: the channels-last bottleneck tensor read in the
// reference's (c, z, y, x) flattening order) made every lane gather 2-byte elements pc apart; when its fp32 image fits the LDS (s_x, [NB][k_in]) the workgroup reads x
// once in PHYSICAL order (coalesced) and writes it at its logical index, and the k loop reads consecutive words.
template <int NB>
__device__ __forceinline__ void linear_fwd_body(const void* __restrict__ x, int x_dtype, const float* __restrict__ w,
                                                const float* __restrict__ bias, float* __restrict__ y, int batch, int k_in,
                                                int j_out, int pc, int pv, int relu, float* s_x) {
    const int j = blockIdx.x;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    const float* wr = w + (size_t)j * k_in;
    const int nt = (int)blockDim.x;                       // 256, or 1024 for the wide layers (one round of loads instead of several)
    // four k per lane: every row of w and x 16-byte aligned, and (permuted x) the four k of a group in one channel (pv % 4 == 0)
    const bool vec = (k_in & 3) == 0 && ((uintptr_t)w & 15) == 0 && (pc == 0 ? ((uintptr_t)x & 15) == 0 : (pv & 3) == 0);
    if (vec) {
        constexpr int LU4 = NB <= 1 ? 8 : (NB <= 2 ? 4 : (NB <= 4 ? 4 : (NB <= 8 ? 2 : 1)));      // 1024-thread workgroups: 128 VGPRs (LU4 * NB * 4 activation values in flight)
        const int nk4 = k_in >> 2;
        const f32x4* wr4 = (const f32x4*)wr;
        if (s_x != nullptr) {                             // workgroup-uniform
            for (int qi = threadIdx.x; qi < k_in; qi += nt) {
                const int v = qi / pc, c = qi - v * pc;   // physical index qi = v * pc + c  <->  logical k = c * pv + v (phys_index's inverse)
#pragma unroll
                for (int b = 0; b < NB; ++b) s_x[b * k_in + c * pv + v] = ld_any(x, x_dtype, (long long)(b < batch ? b : batch - 1) * k_in + qi);
            }
            __syncthreads();
        }
        for (int k0 = threadIdx.x; k0 < nk4; k0 += nt * LU4) {
            f32x4 wv[LU4];
            float xv[LU4][NB][4];
#pragma unroll
            for (int u = 0; u < LU4; ++u) {
                const int k4 = k0 + u * nt;
                const bool ok = k4 < nk4;
                wv[u] = ok ? wr4[k4] : f32x4{0.f, 0.f, 0.f, 0.f};
                const int kk = 4 * (ok ? k4 : 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const long long row = (long long)(b < batch ? b : batch - 1) * k_in;
                    if (s_x != nullptr) {
                        const f32x4 t = *(const f32x4*)(s_x + b * k_in + kk);
                        xv[u][b][0] = t[0]; xv[u][b][1] = t[1]; xv[u][b][2] = t[2]; xv[u][b][3] = t[3];
                    } else if (pc > 0) {
                        const long long ph = phys_index(kk, pc, pv);              // k .. k + 3: the same channel, voxels v .. v + 3
#pragma unroll
                        for (int i = 0; i < 4; ++i) xv[u][b][i] = ld_any(x, x_dtype, row + ph + (long long)i * pc);
                    } else if (x_dtype == VS_F32) {
                        const f32x4 t = *(const f32x4*)((const float*)x + row + kk);
                        xv[u][b][0] = t[0]; xv[u][b][1] = t[1]; xv[u][b][2] = t[2]; xv[u][b][3] = t[3];
                    } else {
                        const u32x2 t = *(const u32x2*)((const unsigned short*)x + row + kk);
                        if (x_dtype == VS_BF16) {
                            xv[u][b][0] = H16<unsigned short>::lo(t[0]); xv[u][b][1] = H16<unsigned short>::hi(t[0]);
                            xv[u][b][2] = H16<unsigned short>::lo(t[1]); xv[u][b][3] = H16<unsigned short>::hi(t[1]);
                        } else {
                            xv[u][b][0] = H16<vs_half>::lo(t[0]); xv[u][b][1] = H16<vs_half>::hi(t[0]);
                            xv[u][b][2] = H16<vs_half>::lo(t[1]); xv[u][b][3] = H16<vs_half>::hi(t[1]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LU4; ++u)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[b] += wv[u][i] * xv[u][b][i];
        }
    } else {
        // one k per lane and load (ragged widths, pv % 4 != 0): LU k-steps per round
        constexpr int LU = NB <= 2 ? 16 : (NB <= 4 ? 8 : 4);
        for (int k0 = threadIdx.x; k0 < k_in; k0 += nt * LU) {
            float wv[LU], xv[LU][NB];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int k = k0 + u * nt;
                const bool ok = k < k_in;
                wv[u] = ok ? wr[k] : 0.f;
                const long long ph = phys_index(ok ? k : 0, pc, pv);
#pragma unroll
                for (int b = 0; b < NB; ++b) xv[u][b] = ld_any(x, x_dtype, (long long)(b < batch ? b : batch - 1) * k_in + ph);
            }
#pragma unroll
            for (int u = 0; u < LU; ++u)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] += wv[u] * xv[u][b];
        }
    }
    __shared__ float red[16][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float s = wave_sum(acc[b]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][b] = s;
    }
    __syncthreads();
    if (threadIdx.x < batch) {
        float s = bias ? bias[j] : 0.f;
        for (int w2 = 0; w2 < (nt >> 6); ++w2) s += red[w2][threadIdx.x];
        if (relu && s < 0.f) s = 0.f;
        y[(size_t)threadIdx.x * j_out + j] = s;
    }
}
template <int NB>
__global__ __launch_bounds__(1024) void linear_fwd_kernel(const void* __restrict__ x, int x_dtype, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int batch, int k_in,
                                                         int j_out, int pc, int pv, int relu, int lds_x) {
    extern __shared__ __attribute__((aligned(16))) float s_lin[];
    linear_fwd_body<NB>(x, x_dtype, w, bias, y, batch, k_in, j_out, pc, pv, relu, lds_x ? s_lin : nullptr);
}
// fc_mean and fc_std read the same bottleneck activation (joint_model.py:241-243): blockIdx.y picks the layer, one launch for both
struct LinearPair { const float* w[2]; const float* bias[2]; float* y[2]; int relu[2]; };
template <int NB>
__global__ __launch_bounds__(1024) void linear_fwd_pair_kernel(const void* __restrict__ x, int x_dtype, const LinearPair a, int batch, int k_in,
                                                              int j_out, int pc, int pv, int lds_x) {
    extern __shared__ __attribute__((aligned(16))) float s_lin[];
    const int op = blockIdx.y;
    linear_fwd_body<NB>(x, x_dtype, a.w[op], a.bias[op], a.y[op], batch, k_in, j_out, pc, pv, a.relu[op], lds_x ? s_lin : nullptr);
}
// bytes of the LDS image of x (0: read it from memory as before — not permuted, or too large for the default 64 KB)
static inline size_t lin_lds_bytes(int nb, int k_in, int pc, int pv) {
    const size_t need = (size_t)nb * k_in * sizeof(float);
    return (pc > 0 && (k_in & 3) == 0 && (pv & 3) == 0 && need <= 64 * 1024) ? need : 0;      // only the four-k-per-lane path stages x
}

extern "C" int vs_linear_fwd(const void* x, int x_dtype, const float* wgt, const float* bias, float* y, int batch, int k_in,
                             int j_out, int pc, int pv, int relu, void* stream) {
    if (!x || !wgt || !y || batch <= 0 || batch > LIN_MAXB || k_in <= 0 || j_out <= 0) return VS_EINVAL;
    if (pc > 0 && (long long)pc * pv != k_in) return VS_ESHAPE;
    const int lin_threads = k_in >= 4096 ? 1024 : 256;
#define LIN_FWD(NB) hipLaunchKernelGGL(linear_fwd_kernel<NB>, dim3(j_out), dim3(lin_threads), lin_lds_bytes(NB, k_in, pc, pv), (hipStream_t)stream, x, x_dtype, wgt, bias, y, batch, k_in, j_out, pc, pv, relu, lin_lds_bytes(NB, k_in, pc, pv) ? 1 : 0)
    if (batch == 1) LIN_FWD(1); else if (batch == 2) LIN_FWD(2); else if (batch <= 4) LIN_FWD(4); else if (batch <= 8) LIN_FWD(8); else LIN_FWD(16);
#undef LIN_FWD
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_linear_fwd_pair(const void* x, int x_dtype, const float* w1, const float* b1, float* y1, int relu1, const float* w2,
                                  const float* b2, float* y2, int relu2, int batch, int k_in, int j_out, int pc, int pv, void* stream) {
    if (!x || !w1 || !w2 || !y1 || !y2 || batch <= 0 || batch > LIN_MAXB || k_in <= 0 || j_out <= 0) return VS_EINVAL;
    if (pc > 0 && (long long)pc * pv != k_in) return VS_ESHAPE;
    LinearPair a;
    a.w[0] = w1; a.w[1] = w2; a.bias[0] = b1; a.bias[1] = b2; a.y[0] = y1; a.y[1] = y2; a.relu[0] = relu1; a.relu[1] = relu2;
    const int lin_threads = k_in >= 4096 ? 1024 : 256;
#define LIN_FWD2(NB) hipLaunchKernelGGL(linear_fwd_pair_kernel<NB>, dim3(j_out, 2), dim3(lin_threads), lin_lds_bytes(NB, k_in, pc, pv), (hipStream_t)stream, x, x_dtype, a, batch, k_in, j_out, pc, pv, lin_lds_bytes(NB, k_in, pc, pv) ? 1 : 0)
    if (batch == 1) LIN_FWD2(1); else if (batch == 2) LIN_FWD2(2); else if (batch <= 4) LIN_FWD2(4); else if (batch <= 8) LIN_FWD2(8); else LIN_FWD2(16);
#undef LIN_FWD2
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// one wave per output j (k_in is small, e.g. 128): y[b][phys(j)] = bias[j] + sum_k W[j][k] z[b][k]
__global__ __launch_bounds__(256) void linear_fwd_perm_out_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, void* __restrict__ y, int y_dtype,
                                                                  int batch, int k_in, int j_out, int pc, int pv) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= j_out) return;
    const float* wr = w + (size_t)j * k_in;
    const long long ph = phys_index(j, pc, pv);
    for (int b = 0; b < batch; ++b) {
        float s = 0.f;
        for (int k = lane; k < k_in; k += 64) s += wr[k] * z[(size_t)b * k_in + k];
        s = wave_sum(s);
        if (lane == 0) st_any(y, y_dtype, (long long)b * j_out + ph, s + (bias ? bias[j] : 0.f));
    }
}

extern "C" int vs_linear_fwd_perm_out(const float* z, const float* wgt, const float* bias, void* y, int y_dtype, int batch,
                                      int k_in, int j_out, int pc, int pv, void* stream) {
    if (!z || !wgt || !y || batch <= 0 || k_in <= 0 || j_out <= 0) return VS_EINVAL;
    if (pc > 0 && (long long)pc * pv != j_out) return VS_ESHAPE;
    hipLaunchKernelGGL(linear_fwd_perm_out_kernel, dim3((j_out + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, wgt, bias, y, y_dtype, batch, k_in, j_out, pc, pv);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// gx[b][phys(k)] = sum_j gy'[b][j] W[j][k]   (gy' = gy masked by y>0 when y_for_relu given).  A workgroup = 64 values of k x 4 waves; wave w walks the
// weight rows j = w, w + 4, ... with 8 rows in flight per step and the waves' partial sums meet in LDS (fixed order).  (One thread per k walking all j_out rows
// — 16 dependent rounds of 8 loads on 108 waves — was 11.7 us for 3.5 MB of weights; before that, a dependent load per FMA: 64 us.)
template <int NB>
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ gy, const float* __restrict__ yrelu,
                                                          void* __restrict__ gx, int x_dtype, int batch, int k_in, int j_out, int pc, int pv) {
    __shared__ float red[3][NB][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int k = blockIdx.x * 64 + lane;
    const bool kok = k < k_in;
    float s[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = 0.f;
    for (int j0 = wave; j0 < j_out; j0 += 32) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = (kok && j0 + 4 * u < j_out) ? w[(size_t)(j0 + 4 * u) * k_in + k] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u < j_out ? j0 + 4 * u : j_out - 1;     // rows past the end carry a zero weight
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int bb = b < batch ? b : batch - 1;
                float g = gy[(size_t)bb * j_out + j];
                if (yrelu && !(yrelu[(size_t)bb * j_out + j] > 0.f)) g = 0.f;
                s[b] += g * wv[u];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) red[wave - 1][b][lane] = s[b];
    }
    __syncthreads();
    if (wave == 0 && kok) {
        const long long ph = phys_index(k, pc, pv);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < batch) st_any(gx, x_dtype, (long long)b * k_in + ph, ((s[b] + red[0][b][lane]) + red[1][b][lane]) + red[2][b][lane]);
    }
}
// gw[j][k] = sum_b gy'[b][j] x[b][phys(k)] ; gb[j] = sum_b gy'[b][j]
__global__ void linear_bwd_w_kernel(const void* __restrict__ x, int x_dtype, const float* __restrict__ gy, const float* __restrict__ yrelu,
                                    float* __restrict__ gw, float* __restrict__ gb, int batch, int k_in, int j_out, int pc, int pv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)j_out * k_in) return;
    const int j = (int)(i / k_in), k = (int)(i - (long long)j * k_in);
    const long long ph = phys_index(k, pc, pv);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < batch; ++b) {
        float g = gy[(size_t)b * j_out + j];
        if (yrelu && !(yrelu[(size_t)b * j_out + j] > 0.f)) g = 0.f;
        s += g * ld_any(x, x_dtype, (long long)b * k_in + ph);
        sb += g;
    }
    if (gw) gw[i] = s;
    if (gb && k == 0) gb[j] = sb;
}

extern "C" int vs_linear_bwd(const void* x, int x_dtype, const float* wgt, const float* gy, const float* y_for_relu, void* gx,
                             float* gw, float* gb, int batch, int k_in, int j_out, int pc, int pv, void* stream) {
    if (!wgt || !gy || batch <= 0 || k_in <= 0 || j_out <= 0) return VS_EINVAL;
    if (pc > 0 && (long long)pc * pv != k_in) return VS_ESHAPE;
    if (gx) {
        if (batch > LIN_MAXB) return VS_EINVAL;
#define LIN_BX(NB) hipLaunchKernelGGL(linear_bwd_x_kernel<NB>, dim3((k_in + 63) / 64), dim3(256), 0, (hipStream_t)stream, wgt, gy, y_for_relu, gx, x_dtype, batch, k_in, j_out, pc, pv)
        if (batch == 1) LIN_BX(1); else if (batch == 2) LIN_BX(2); else if (batch <= 4) LIN_BX(4); else if (batch <= 8) LIN_BX(8); else LIN_BX(16);
#undef LIN_BX
        VS_CHECK_LAUNCH();
    }
    if (gw || gb) {
        if (!x) return VS_EINVAL;
        const long long total = (long long)j_out * k_in;
        hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, gy, y_for_relu, gw, gb, batch, k_in, j_out, pc, pv);
        VS_CHECK_LAUNCH();
    }
    return VS_OK;
}

// backward of y[b][phys(j)] = bias[j] + sum_k W[j][k] z[b][k]
__global__ __launch_bounds__(256) void linear_perm_out_bwd_z_kernel(const float* __restrict__ w, const void* __restrict__ gy, int y_dtype,
                                                                    float* __restrict__ gz, int batch, int k_in, int j_out, int pc, int pv) {
    // one workgroup per (b, k): reduce over j
    const int b = blockIdx.y, k = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < j_out; j += 256)
        s += ld_any(gy, y_dtype, (long long)b * j_out + phys_index(j, pc, pv)) * w[(size_t)j * k_in + k];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) gz[(size_t)b * k_in + k] = red[0] + red[1] + red[2] + red[3];
}
__global__ void linear_perm_out_bwd_w_kernel(const float* __restrict__ z, const void* __restrict__ gy, int y_dtype, float* __restrict__ gw,
                                             float* __restrict__ gb, int batch, int k_in, int j_out, int pc, int pv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)j_out * k_in) return;
    const int j = (int)(i / k_in), k = (int)(i - (long long)j * k_in);
    const long long ph = phys_index(j, pc, pv);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < batch; ++b) {
        const float g = ld_any(gy, y_dtype, (long long)b * j_out + ph);
        s += g * z[(size_t)b * k_in + k];
        sb += g;
    }
    if (gw) gw[i] = s;
    if (gb && k == 0) gb[j] = sb;
}
extern "C" int vs_linear_perm_out_bwd(const float* z, const float* wgt, const void* gy, int y_dtype, float* gz, float* gw,
                                      float* gb, int batch, int k_in, int j_out, int pc, int pv, void* stream) {
    if (!wgt || !gy || batch <= 0 || k_in <= 0 || j_out <= 0) return VS_EINVAL;
    if (pc > 0 && (long long)pc * pv != j_out) return VS_ESHAPE;
    if (gz) {
        hipLaunchKernelGGL(linear_perm_out_bwd_z_kernel, dim3(k_in, batch), dim3(256), 0, (hipStream_t)stream, wgt, gy, y_dtype, gz, batch, k_in, j_out, pc, pv);
        VS_CHECK_LAUNCH();
    }
    if (gw || gb) {
        if (!z) return VS_EINVAL;
        const long long total = (long long)j_out * k_in;
        hipLaunchKernelGGL(linear_perm_out_bwd_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, gy, y_dtype, gw, gb, batch, k_in, j_out, pc, pv);
        VS_CHECK_LAUNCH();
    }
    return VS_OK;
}

// ---- reparameterisation / KL -----------------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float* mean, const float* sd, const float* noise, float scale, float* z, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = mean[i] + noise[i] * sd[i] * scale;
}
__global__ void reparam_bwd_kernel(const float* gz, const float* noise, float scale, float* gmean, float* gstd, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (gmean) gmean[i] = gz[i];
        if (gstd) gstd[i] = gz[i] * noise[i] * scale;
    }
}
extern "C" int vs_reparam_fwd(const float* mean, const float* std_, const float* noise, float scale, float* z, long long count, void* stream) {
    if (!mean || !std_ || !noise || !z || count <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(reparam_fwd_kernel, GRID1D(count), dim3(256), 0, (hipStream_t)stream, mean, std_, noise, scale, z, count);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
extern "C" int vs_reparam_bwd(const float* gz, const float* noise, float scale, float* gmean, float* gstd, long long count, void* stream) {
    if (!gz || !noise || count <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(reparam_bwd_kernel, GRID1D(count), dim3(256), 0, (hipStream_t)stream, gz, noise, scale, gmean, gstd, count);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ sd, float* __restrict__ out, int batch, int dim) {
    // single workgroup: total = sum_b 0.5*(sum sd^2 + sum mean^2 - 2 sum log(sd+1e-5)) / batch
    double s = 0.0;
    for (int i = threadIdx.x; i < batch * dim; i += 256) {
        const float m = mean[i], d = sd[i];
        s += 0.5 * ((double)d * d + (double)m * m - 2.0 * (double)logf(d + 0.00001f));
    }
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / batch);
}
__global__ void kl_bwd_kernel(const float* mean, const float* sd, const float* gout, float* gmean, float* gstd, int batch, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * dim) return;
    const float g = gout[0] / batch;
    if (gmean) gmean[i] = g * mean[i];
    if (gstd) gstd[i] = g * (sd[i] - 1.f / (sd[i] + 0.00001f));
}
extern "C" int vs_kl_fwd(const float* mean, const float* std_, float* out, int batch, int dim, void* stream) {
    if (!mean || !std_ || !out || batch <= 0 || dim <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, std_, out, batch, dim);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
extern "C" int vs_kl_bwd(const float* mean, const float* std_, const float* gout, float* gmean, float* gstd, int batch, int dim, void* stream) {
    if (!mean || !std_ || !gout || batch <= 0 || dim <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((batch * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, std_, gout, gmean, gstd, batch, dim);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- soft Dice ------------------------------------------------------------------------------------------
// grid (blocks, channel slot, b): float4 streaming over the voxels of one (b,c) plane; fp64 atomics per block.
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ s, const float* __restrict__ t, double* __restrict__ sums,
                                                        int channels, long long voxels, int bot) {
    const int c = bot + blockIdx.y, b = blockIdx.z;
    const float* sp = s + ((size_t)b * channels + c) * voxels;
    const float* tp = t + ((size_t)b * channels + c) * voxels;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const long long v4 = voxels / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
        const f32x4 x = *(const f32x4*)(sp + i * 4), y = *(const f32x4*)(tp + i * 4);
        a0 += (double)(x[0] * y[0] + x[1] * y[1]) + (double)(x[2] * y[2] + x[3] * y[3]);
        a1 += (double)(x[0] + x[1]) + (double)(x[2] + x[3]);
        a2 += (double)(y[0] + y[1]) + (double)(y[2] + y[3]);
    }
    if (blockIdx.x == 0)
        for (long long i = v4 * 4 + threadIdx.x; i < voxels; i += 256) { a0 += (double)sp[i] * tp[i]; a1 += sp[i]; a2 += tp[i]; }
    __shared__ double red[4][3];
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a0; red[threadIdx.x >> 6][1] = a1; red[threadIdx.x >> 6][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(sums + ((size_t)b * channels + c) * 3 + threadIdx.x, tot);
    }
}
__global__ void dice_finish_kernel(const double* __restrict__ sums, float* per_sample, float* mean_out, int batch, int channels, int bot, int top, float eps) {
    // single small block; thread b computes per-sample mean over channels
    __shared__ float ps[64];
    const int b = threadIdx.x;
    float v = 0.f;
    if (b < batch) {
        for (int c = bot; c < top; ++c) {
            const double* q = sums + ((size_t)b * channels + c) * 3;
            // fp32 arithmetic as torch does on the fp32 sums
            const float I = (float)q[0], S = (float)q[1], Tt = (float)q[2];
            v += 2.f * I / (S + Tt + eps);
        }
        v /= (float)(top - bot);
        if (per_sample) per_sample[b] = v;
        ps[b] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0 && mean_out) {
        float m = 0.f;
        for (int i = 0; i < batch; ++i) m += ps[i];
        mean_out[0] = m / batch;
    }
}
extern "C" int vs_dice_fwd(const float* s, const float* t, double* sums, float* per_sample, float* mean_out, int batch,
                           int channels, long long voxels, int bot, int top, float eps, void* stream) {
    if (!s || !t || !sums || batch <= 0 || batch > 64 || channels <= 0 || voxels <= 0 || bot < 0 || top > channels || bot >= top) return VS_EINVAL;
    if (((uintptr_t)s & 15) || ((uintptr_t)t & 15) || (voxels & 3)) return VS_EALIGN;
    hipError_t e = vs_zero_async(sums, sizeof(double) * 3 * batch * channels, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    long long blocks = (voxels / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    if (VS_DET_BUILD) blocks = 1;                      // deterministic mode: one block per (b, c) plane, so one fp64 atomic onto a zeroed word
    hipLaunchKernelGGL(dice_sums_kernel, dim3((unsigned)blocks, top - bot, batch), dim3(256), 0, (hipStream_t)stream, s, t, sums, channels, voxels, bot);
    VS_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, per_sample, mean_out, batch, channels, bot, top, eps);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// d/ds_i [2I/(S+T+eps)] = 2 t_i/den - 2I/den^2 ; d/dt_i symmetric.  grid (blocks, channels, b)
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t, const double* __restrict__ sums,
                                                       const float* __restrict__ gout, int gout_is_mean, int batch,
                                                       float* __restrict__ gs, float* __restrict__ gt,
                                                       int channels, long long voxels, int bot, int top, float eps) {
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t plane = ((size_t)b * channels + c) * voxels;
    const long long v4 = voxels / 4;
    if (c < bot || c >= top) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
            if (gs) *(f32x4*)(gs + plane + i * 4) = z;
            if (gt) *(f32x4*)(gt + plane + i * 4) = z;
        }
        return;
    }
    const double* q = sums + ((size_t)b * channels + c) * 3;
    const float I = (float)q[0], den = (float)q[1] + (float)q[2] + eps;
    const float g = (gout_is_mean ? gout[0] / (float)batch : gout[b]) / (float)(top - bot);
    const float k1 = g * 2.f / den, k2 = g * 2.f * I / (den * den);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
        if (gs) { const f32x4 y = *(const f32x4*)(t + plane + i * 4); *(f32x4*)(gs + plane + i * 4) = f32x4{k1 * y[0] - k2, k1 * y[1] - k2, k1 * y[2] - k2, k1 * y[3] - k2}; }
        if (gt) { const f32x4 x = *(const f32x4*)(s + plane + i * 4); *(f32x4*)(gt + plane + i * 4) = f32x4{k1 * x[0] - k2, k1 * x[1] - k2, k1 * x[2] - k2, k1 * x[3] - k2}; }
    }
}
extern "C" int vs_dice_bwd(const float* s, const float* t, const double* sums, const float* gout, int gout_is_mean, float* gs,
                           float* gt, int batch, int channels, long long voxels, int bot, int top, float eps, void* stream) {
    if (!s || !t || !sums || !gout || batch <= 0 || channels <= 0 || voxels <= 0 || (voxels & 3)) return VS_EINVAL;
    long long blocks = (voxels / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)blocks, channels, batch), dim3(256), 0, (hipStream_t)stream, s, t, sums, gout, gout_is_mean, batch, gs, gt, channels, voxels, bot, top, eps);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- weighted sum of Dice losses, one source against k targets (one launch forward, one backward) -----------------
#define DICE_MULTI_MAX 4
struct DiceMulti {
    const float* s;
    const float* t[DICE_MULTI_MAX];
    const float* lab[DICE_MULTI_MAX];      // non-null: target j is the one-hot of these labels ([B][voxels] floats, truncated as vs_onehot does) — never materialised
    float* gt[DICE_MULTI_MAX];
    float w[DICE_MULTI_MAX];
    int k;
};
// scratch layout: sums[(j*B + b)*C + c][3] = (I_j, S, T_j), written by the finish kernel (the backward reads them); then
// the per-block partials part[((b*nc + ci)*NQ + q)*nblk + blk], q = 0: S, 1+2j: I_j, 2+2j: T_j.  No atomics: 27 ns per
// same-line fp64 atomic made a 432-block single-launch version (atomics + last-block ticket) cost 58 us; two small launches
// cost 12, and the fixed summation order makes the loss bitwise reproducible.
#define DICE_MULTI_BLOCKS 256
// four consecutive values of target j in channel c of sample b
__device__ __forceinline__ f32x4 dice_target4(const DiceMulti& a, int j, size_t plane, int b, int c, long long voxels, long long i) {
    if (a.lab[j] != nullptr) {
        const f32x4 l = *(const f32x4*)(a.lab[j] + (size_t)b * voxels + i * 4);
        return f32x4{(int)l[0] == c ? 1.f : 0.f, (int)l[1] == c ? 1.f : 0.f, (int)l[2] == c ? 1.f : 0.f, (int)l[3] == c ? 1.f : 0.f};
    }
    return *(const f32x4*)(a.t[j] + plane + i * 4);
}
__global__ __launch_bounds__(256) void dice_multi_partial_kernel(const DiceMulti a, double* __restrict__ part, int channels, long long voxels, int bot) {
    const int c = bot + blockIdx.y, b = blockIdx.z, K = a.k;
    const size_t plane = ((size_t)b * channels + c) * voxels;
    const float* sp = a.s + plane;
    double aS = 0.0, aI[DICE_MULTI_MAX], aT[DICE_MULTI_MAX];
#pragma unroll
    for (int j = 0; j < DICE_MULTI_MAX; ++j) { aI[j] = 0.0; aT[j] = 0.0; }
    const long long v4 = voxels / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
        const f32x4 x = *(const f32x4*)(sp + i * 4);
        aS += (double)(x[0] + x[1]) + (double)(x[2] + x[3]);
#pragma unroll
        for (int j = 0; j < DICE_MULTI_MAX; ++j) {
            if (j < K) {
                const f32x4 y = dice_target4(a, j, plane, b, c, voxels, i);
                aI[j] += (double)(x[0] * y[0] + x[1] * y[1]) + (double)(x[2] * y[2] + x[3] * y[3]);
                aT[j] += (double)(y[0] + y[1]) + (double)(y[2] + y[3]);
            }
        }
    }
    __shared__ double red[4][1 + 2 * DICE_MULTI_MAX];
    const int wave = threadIdx.x >> 6;
    aS = wave_sum_d(aS);
#pragma unroll
    for (int j = 0; j < DICE_MULTI_MAX; ++j) { aI[j] = wave_sum_d(aI[j]); aT[j] = wave_sum_d(aT[j]); }
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = aS;
#pragma unroll
        for (int j = 0; j < DICE_MULTI_MAX; ++j) { red[wave][1 + 2 * j] = aI[j]; red[wave][2 + 2 * j] = aT[j]; }
    }
    __syncthreads();
    const int NQ = 1 + 2 * K;
    if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        part[(((size_t)b * gridDim.y + blockIdx.y) * NQ + q) * gridDim.x + blockIdx.x] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
    }
}

__global__ __launch_bounds__(256) void dice_multi_finish_kernel(const DiceMulti a, const double* __restrict__ part, double* __restrict__ sums,
                                                               float* __restrict__ terms, float* __restrict__ final_out, int batch,
                                                               int channels, int bot, int top, int nblk, float eps) {
    const int K = a.k, NQ = 1 + 2 * K, nc = top - bot;
    __shared__ double s_tot[64 * 2 * (1 + 2 * DICE_MULTI_MAX)];      // [b][ci][q], batch*nc <= 128 checked on the host
    __shared__ double s_p16[16][17];
    // 16 threads per statistic, each summing every 16th block (fixed order: reproducible), 16 statistics per round: the usual 10
    // statistics x 256 blocks are two rounds of 8 loads per thread instead of 32 dependent rounds in one thread
    const int sl = threadIdx.x & 15, sg = threadIdx.x >> 4;
    for (int i0 = 0; i0 < batch * nc * NQ; i0 += 16) {
        const int i = i0 + sg;
        double t = 0.0;
        if (i < batch * nc * NQ) {
            const double* src = part + (size_t)i * nblk;
            int blk = sl;
            for (; blk + 7 * 16 < nblk; blk += 8 * 16) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[blk + 16 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; blk < nblk; blk += 16) t += src[blk];
        }
        s_p16[sg][sl] = t;
        __syncthreads();
        if (sl == 0 && i < batch * nc * NQ) {
            double tt = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) tt += s_p16[sg][u];
            s_tot[i] = tt;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < batch * nc * NQ; i += 256) {
        const double t = s_tot[i];
        const int q = i % NQ, bc = i / NQ, ci = bc % nc, b = bc / nc;
        if (q == 0) {
            for (int j = 0; j < K; ++j) sums[(((size_t)j * batch + b) * channels + bot + ci) * 3 + 1] = t;
        } else {
            const int j = (q - 1) >> 1;
            sums[(((size_t)j * batch + b) * channels + bot + ci) * 3 + (((q - 1) & 1) ? 2 : 0)] = t;
        }
    }
    __syncthreads();
    __shared__ float s_term[DICE_MULTI_MAX];
    if (threadIdx.x < K) {
        const int j = threadIdx.x;
        float m = 0.f;
        for (int b = 0; b < batch; ++b) {
            float v = 0.f;
            for (int ci = 0; ci < nc; ++ci) {
                const double* q = s_tot + (b * nc + ci) * NQ;
                // fp32 arithmetic as torch does on the fp32 sums
                const float I = (float)q[1 + 2 * j], S = (float)q[0], T = (float)q[2 + 2 * j];
                v += 2.f * I / (S + T + eps);
            }
            m += v / (float)nc;
        }
        const float term = 1.f - m / (float)batch;
        terms[j] = term;
        s_term[j] = term;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float f = __fmul_rn(a.w[0], s_term[0]);
        for (int j = 1; j < K; ++j) f = __fadd_rn(f, __fmul_rn(a.w[j], s_term[j]));
        final_out[0] = f;
    }
}

static int dice_multi_args(DiceMulti& a, const float* s, const float* const* t, const float* w, float* const* gt, int k, const float* const* lab = nullptr) {
    if (!s || !t || !w || k <= 0 || k > DICE_MULTI_MAX) return VS_EINVAL;
    a = DiceMulti{};
    a.s = s; a.k = k;
    if ((uintptr_t)s & 15) return VS_EALIGN;
    for (int j = 0; j < k; ++j) {
        const float* lj = lab ? lab[j] : nullptr;
        if (!t[j] && !lj) return VS_EINVAL;
        if (lj && gt && gt[j]) return VS_EINVAL;                 // a label target has no gradient
        if (((uintptr_t)t[j] & 15) || ((uintptr_t)lj & 15) || (gt && gt[j] && ((uintptr_t)gt[j] & 15))) return VS_EALIGN;
        a.t[j] = lj ? nullptr : t[j]; a.lab[j] = lj; a.w[j] = w[j]; a.gt[j] = gt ? gt[j] : nullptr;
    }
    return VS_OK;
}

extern "C" size_t vs_dice_loss_multi_scratch_doubles(int k, int batch, int channels) {
    if (k <= 0 || batch <= 0 || channels <= 0) return 0;
    return (size_t)3 * k * batch * channels + (size_t)batch * channels * (1 + 2 * k) * DICE_MULTI_BLOCKS;
}

extern "C" int vs_dice_loss_multi_fwd(const float* s, const float* const* t, const float* w, int k, double* scratch, float* terms,
                                      float* final_out, int batch, int channels, long long voxels, int bot, int top, float eps, void* stream) {
    return vs_dice_loss_multi_labels_fwd(s, t, nullptr, w, k, scratch, terms, final_out, batch, channels, voxels, bot, top, eps, stream);
}

extern "C" int vs_dice_loss_multi_labels_fwd(const float* s, const float* const* t, const float* const* labels, const float* w, int k, double* scratch,
                                             float* terms, float* final_out, int batch, int channels, long long voxels, int bot, int top, float eps,
                                             void* stream) {
    DiceMulti a;
    int rc = dice_multi_args(a, s, t, w, nullptr, k, labels);
    if (rc) return rc;
    if (!scratch || !terms || !final_out || batch <= 0 || batch > 64 || channels <= 0 || voxels <= 0 || bot < 0 || top > channels || bot >= top) return VS_EINVAL;
    if (voxels & 3) return VS_EALIGN;
    if ((long long)batch * (top - bot) > 128) return VS_ESHAPE;
    long long blocks = (voxels / 4 + 256 * 2 - 1) / (256 * 2);       // short loops: the launch is latency-, not bandwidth-bound
    if (blocks < 1) blocks = 1;
    if (blocks > DICE_MULTI_BLOCKS) blocks = DICE_MULTI_BLOCKS;
    double* part = scratch + (size_t)3 * k * batch * channels;
    hipLaunchKernelGGL(dice_multi_partial_kernel, dim3((unsigned)blocks, top - bot, batch), dim3(256), 0, (hipStream_t)stream, a, part, channels,
                       voxels, bot);
    VS_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_multi_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, part, scratch, terms, final_out, batch, channels,
                       bot, top, (int)blocks, eps);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// final = sum_j w_j (1 - mean_{b,c} 2 I_j/(S+T_j+eps))  =>  d/ds_i = -sum_j w_j/(B*nc) (2 t_ji/den_j - 2 I_j/den_j^2), d/dt_ji symmetric
__global__ __launch_bounds__(256) void dice_multi_bwd_kernel(const DiceMulti a, const double* __restrict__ sums, const float* __restrict__ gout,
                                                            float* __restrict__ gs, int batch, int channels, long long voxels, int bot,
                                                            int top, float eps) {
    const int c = blockIdx.y, b = blockIdx.z, K = a.k;
    const size_t plane = ((size_t)b * channels + c) * voxels;
    const long long v4 = voxels / 4;
    if (c < bot || c >= top) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
            if (gs) *(f32x4*)(gs + plane + i * 4) = z;
#pragma unroll
            for (int j = 0; j < DICE_MULTI_MAX; ++j)
                if (j < K && a.gt[j]) *(f32x4*)(a.gt[j] + plane + i * 4) = z;
        }
        return;
    }
    float k1[DICE_MULTI_MAX], k2[DICE_MULTI_MAX], k2sum = 0.f;
    const float g0 = -gout[0] / ((float)batch * (float)(top - bot));
#pragma unroll
    for (int j = 0; j < DICE_MULTI_MAX; ++j) {
        k1[j] = 0.f; k2[j] = 0.f;
        if (j < K) {
            const double* q = sums + (((size_t)j * batch + b) * channels + c) * 3;
            const float I = (float)q[0], den = (float)q[1] + (float)q[2] + eps;
            const float g = g0 * a.w[j];
            k1[j] = g * 2.f / den;
            k2[j] = g * 2.f * I / (den * den);
            k2sum += k2[j];
        }
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v4; i += (long long)gridDim.x * 256) {
        const f32x4 x = *(const f32x4*)(a.s + plane + i * 4);
        f32x4 acc = f32x4{-k2sum, -k2sum, -k2sum, -k2sum};
#pragma unroll
        for (int j = 0; j < DICE_MULTI_MAX; ++j) {
            if (j < K) {
                if (gs) {
                    const f32x4 y = dice_target4(a, j, plane, b, c, voxels, i);
                    acc[0] += k1[j] * y[0]; acc[1] += k1[j] * y[1]; acc[2] += k1[j] * y[2]; acc[3] += k1[j] * y[3];
                }
                if (a.gt[j]) *(f32x4*)(a.gt[j] + plane + i * 4) = f32x4{k1[j] * x[0] - k2[j], k1[j] * x[1] - k2[j], k1[j] * x[2] - k2[j], k1[j] * x[3] - k2[j]};
            }
        }
        if (gs) *(f32x4*)(gs + plane + i * 4) = acc;
    }
}

extern "C" int vs_dice_loss_multi_bwd(const float* s, const float* const* t, const float* w, int k, const double* scratch, const float* gout,
                                      float* gs, float* const* gt, int batch, int channels, long long voxels, int bot, int top, float eps,
                                      void* stream) {
    return vs_dice_loss_multi_labels_bwd(s, t, nullptr, w, k, scratch, gout, gs, gt, batch, channels, voxels, bot, top, eps, stream);
}

extern "C" int vs_dice_loss_multi_labels_bwd(const float* s, const float* const* t, const float* const* labels, const float* w, int k,
                                             const double* scratch, const float* gout, float* gs, float* const* gt, int batch, int channels,
                                             long long voxels, int bot, int top, float eps, void* stream) {
    DiceMulti a;
    int rc = dice_multi_args(a, s, t, w, gt, k, labels);
    if (rc) return rc;
    if (!scratch || !gout || batch <= 0 || channels <= 0 || voxels <= 0 || (voxels & 3) || bot < 0 || top > channels || bot >= top) return VS_EINVAL;
    if (gs && ((uintptr_t)gs & 15)) return VS_EALIGN;
    long long blocks = (voxels / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dice_multi_bwd_kernel, dim3((unsigned)blocks, channels, batch), dim3(256), 0, (hipStream_t)stream, a, scratch, gout, gs,
                       batch, channels, voxels, bot, top, eps);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- BCE ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_sum_kernel(const float* __restrict__ p, const float* __restrict__ t, double* __restrict__ acc, long long count) {
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(logf(1.f - p[i]), -100.f);
        s -= (double)(t[i] * lp + (1.f - t[i]) * lq);
    }
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}
__global__ void bce_finish_kernel(const double* acc, float* out, long long count) { out[0] = (float)(acc[0] / (double)count); }
__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ gout, float* __restrict__ gp, long long count) {
    const float g = gout[0] / (float)count;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float d = fmaxf(p[i] * (1.f - p[i]), 1e-12f);
        gp[i] = g * (p[i] - t[i]) / d;
    }
}
extern "C" int vs_bce_fwd(const float* p, const float* t, float* out, double* scratch, long long count, void* stream) {
    if (!p || !t || !out || !scratch || count <= 0) return VS_EINVAL;
    hipError_t e = vs_zero_async(scratch, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    long long blocks = (count + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    if (VS_DET_BUILD) blocks = 1;                      // deterministic mode: a single block, a single atomic
    hipLaunchKernelGGL(bce_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, t, scratch, count);
    VS_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scratch, out, count);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
extern "C" int vs_bce_bwd(const float* p, const float* t, const float* gout, float* gp, long long count, void* stream) {
    if (!p || !t || !gout || !gp || count <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(bce_bwd_kernel, GRID1D(count), dim3(256), 0, (hipStream_t)stream, p, t, gout, gp, count);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- multi-tensor optimiser steps -----------------------------------------------------------------------------
#define MT_CHUNK 4096
__global__ __launch_bounds__(256) void sgd_multi_kernel(float* const* params, const float* const* grads, float* const* bufs,
                                                        const long long* sizes, const int* block_map, float lr, float momentum,
                                                        float wd, int first, const float* __restrict__ loss_scale,
                                                        const float* __restrict__ found_inf, const float* __restrict__ hyper) {
    if (found_inf != nullptr && found_inf[0] != 0.f) return;          // a non-finite gradient somewhere: the whole step is skipped
    if (hyper != nullptr) { lr = hyper[0]; momentum = hyper[1]; wd = hyper[2]; }      // device-resident hyperparameters: a captured launch follows the schedule
    const float inv_scale = loss_scale != nullptr ? 1.f / loss_scale[0] : 1.f;
    const int ti = block_map[2 * blockIdx.x], start = block_map[2 * blockIdx.x + 1];
    float* p = params[ti];
    const float* g = grads[ti];
    float* m = bufs[ti];
    const long long n = sizes[ti];
    const long long end = (long long)start + MT_CHUNK < n ? (long long)start + MT_CHUNK : n;
    // every operand of the chunk is requested before the first store: the compiler cannot move a load over a store that may alias, and
    // sixteen dependent load -> store rounds made this 2.3 M-parameter update 25 us
    constexpr int R = MT_CHUNK / 256;
    if (end - start == MT_CHUNK && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0)) {      // a full, 16-byte-aligned chunk (uniform): 12 vector loads instead of 48 dword loads (same 12.5 us: the update moves 49 MB at 3.9 TB/s either way)
        constexpr int R4 = R / 4;
        f32x4 g4[R4], p4[R4], m4[R4];
        const f32x4* gq = (const f32x4*)(g + start);
        f32x4* pq = (f32x4*)(p + start);
        f32x4* mq = (f32x4*)(m + start);
#pragma unroll
        for (int r = 0; r < R4; ++r) {
            const int i = threadIdx.x + r * 256;
            g4[r] = gq[i]; p4[r] = pq[i];
            m4[r] = first ? f32x4{0.f, 0.f, 0.f, 0.f} : mq[i];
        }
#pragma unroll
        for (int r = 0; r < R4; ++r) {
            const int i = threadIdx.x + r * 256;
            f32x4 bo, po;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gi = g4[r][e] * inv_scale + wd * p4[r][e];
                const float bi = first ? gi : momentum * m4[r][e] + gi;
                bo[e] = bi;
                po[e] = p4[r][e] - lr * (momentum != 0.f ? bi : gi);
            }
            mq[i] = bo; pq[i] = po;
        }
        return;
    }
    float gv[R], pv[R], mv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        const bool ok = i < end;
        gv[r] = ok ? g[i] : 0.f; pv[r] = ok ? p[i] : 0.f; mv[r] = (ok && !first) ? m[i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        if (i < end) {
            const float gi = gv[r] * inv_scale + wd * pv[r];
            const float bi = first ? gi : momentum * mv[r] + gi;
            m[i] = bi;
            p[i] = pv[r] - lr * (momentum != 0.f ? bi : gi);
        }
    }
}
extern "C" int vs_sgd_momentum_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                                     const int* block_map, int n_blocks, float lr, float momentum, float weight_decay,
                                     int first_step, void* stream) {
    return vs_sgd_momentum_scaled_multi(params, grads, bufs, sizes, block_map, n_blocks, lr, momentum, weight_decay, first_step, nullptr, nullptr, stream);
}
extern "C" int vs_sgd_momentum_scaled_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                                            const int* block_map, int n_blocks, float lr, float momentum, float weight_decay,
                                            int first_step, const float* loss_scale, const float* found_inf, void* stream) {
    if (!params || !grads || !bufs || !sizes || !block_map || n_blocks <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, params, grads, bufs, sizes, block_map, lr, momentum,
                       weight_decay, first_step, loss_scale, found_inf, (const float*)nullptr);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
extern "C" int vs_sgd_momentum_dev_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                                         const int* block_map, int n_blocks, const float* hyper, const float* loss_scale,
                                         const float* found_inf, void* stream) {
    if (!params || !grads || !bufs || !sizes || !block_map || n_blocks <= 0 || !hyper) return VS_EINVAL;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, params, grads, bufs, sizes, block_map, 0.f, 0.f,
                       0.f, 0, loss_scale, found_inf, hyper);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(float* const* params, const float* const* grads, float* const* m1, float* const* m2,
                                                         const long long* sizes, const int* block_map, float lr, float b1, float b2,
                                                         float omb1, float omb2, float eps, float wd, float bc1, float bc2,
                                                         const float* __restrict__ loss_scale, const float* __restrict__ found_inf) {
    if (found_inf != nullptr && found_inf[0] != 0.f) return;
    const float inv_scale = loss_scale != nullptr ? 1.f / loss_scale[0] : 1.f;
    const int ti = block_map[2 * blockIdx.x], start = block_map[2 * blockIdx.x + 1];
    float* p = params[ti];
    const float* g = grads[ti];
    float* a = m1[ti];
    float* v = m2[ti];
    const long long n = sizes[ti];
    const long long end = (long long)start + MT_CHUNK < n ? (long long)start + MT_CHUNK : n;
    const float step_size = lr / bc1;
    const float bc2s = sqrtf(bc2);
    constexpr int R = MT_CHUNK / 256;                    // all loads of the chunk before the first store (see sgd_multi_kernel)
    float gv[R], pv[R], av[R], vv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        const bool ok = i < end;
        gv[r] = ok ? g[i] : 0.f; pv[r] = ok ? p[i] : 0.f; av[r] = ok ? a[i] : 0.f; vv[r] = ok ? v[i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        if (i < end) {
            const float gi = gv[r] * inv_scale + wd * pv[r];
            const float ai = b1 * av[r] + omb1 * gi;                 // omb = 1 - beta formed in double on the host, as torch forms it
            const float vi = b2 * vv[r] + omb2 * gi * gi;
            a[i] = ai; v[i] = vi;
            const float denom = sqrtf(vi) / bc2s + eps;
            p[i] = pv[r] - step_size * ai / denom;
        }
    }
}
extern "C" int vs_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                             const long long* sizes, const int* block_map, int n_blocks, float lr, double beta1, double beta2,
                             float eps, float weight_decay, int step, void* stream) {
    return vs_adam_scaled_multi(params, grads, exp_avg, exp_avg_sq, sizes, block_map, n_blocks, lr, beta1, beta2, eps, weight_decay, step, nullptr, nullptr, stream);
}
extern "C" int vs_adam_scaled_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                    const long long* sizes, const int* block_map, int n_blocks, float lr, double beta1, double beta2,
                                    float eps, float weight_decay, int step, const float* loss_scale, const float* found_inf, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !block_map || n_blocks <= 0 || step < 1) return VS_EINVAL;
    // betas arrive as doubles (python floats): 1 - beta and the bias corrections are formed in double, as torch.optim.Adam forms them
    const float bc1 = (float)(1.0 - pow(beta1, (double)step)), bc2 = (float)(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes, block_map, lr,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, bc1, bc2, loss_scale, found_inf);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- dynamic loss scaling (fp16 storage: Dice gradients are O(1e-6), below fp16's normal range) ---------------------------------------
__global__ __launch_bounds__(256) void grad_finite_multi_kernel(const float* const* grads, const long long* sizes, const int* block_map,
                                                                float* __restrict__ found_inf) {
    const int ti = block_map[2 * blockIdx.x], start = block_map[2 * blockIdx.x + 1];
    const float* g = grads[ti];
    const long long n = sizes[ti];
    const long long end = (long long)start + MT_CHUNK < n ? (long long)start + MT_CHUNK : n;
    constexpr int R = MT_CHUNK / 256;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        const float v = i < end ? g[i] : 0.f;
        bad |= !(fabsf(v) <= 3.402823466e+38f);         // inf or nan
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;       // same value from every writer: a benign race
}
extern "C" int vs_grad_finite_multi(const float* const* grads, const long long* sizes, const int* block_map, int n_blocks,
                                    float* found_inf, void* stream) {
    if (!grads || !sizes || !block_map || n_blocks <= 0 || !found_inf) return VS_EINVAL;
    hipLaunchKernelGGL(grad_finite_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, grads, sizes, block_map, found_inf);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
__global__ void loss_scale_update_kernel(float* scale, int* growth_tracker, float* found_inf, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (found_inf[0] != 0.f) {
        scale[0] *= backoff;
        growth_tracker[0] = 0;
    } else {
        const int t = growth_tracker[0] + 1;
        if (t >= interval) { scale[0] *= growth; growth_tracker[0] = 0; }
        else growth_tracker[0] = t;
    }
    found_inf[0] = 0.f;
}
extern "C" int vs_loss_scale_update(float* scale, int* growth_tracker, float* found_inf, float growth_factor, float backoff_factor,
                                    int growth_interval, void* stream) {
    if (!scale || !growth_tracker || !found_inf || growth_factor < 1.f || backoff_factor <= 0.f || backoff_factor > 1.f || growth_interval < 1)
        return VS_EINVAL;
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale, growth_tracker, found_inf, growth_factor,
                       backoff_factor, growth_interval);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ __launch_bounds__(256) void ema_multi_kernel(float* const* teacher, const float* const* student, const long long* sizes,
                                                        const int* block_map, float alpha) {
    const int ti = block_map[2 * blockIdx.x], start = block_map[2 * blockIdx.x + 1];
    float* t = teacher[ti];
    const float* s = student[ti];
    const long long n = sizes[ti];
    const long long end = (long long)start + MT_CHUNK < n ? (long long)start + MT_CHUNK : n;
    constexpr int R = MT_CHUNK / 256;                    // all loads of the chunk before the first store (see sgd_multi_kernel)
    float tv[R], sv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        tv[r] = i < end ? t[i] : 0.f; sv[r] = i < end ? s[i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        if (i < end) t[i] = alpha * tv[r] + (1.f - alpha) * sv[r];
    }
}
extern "C" int vs_ema_multi(float* const* teacher, const float* const* student, const long long* sizes, const int* block_map,
                            int n_blocks, float alpha, void* stream) {
    if (!teacher || !student || !sizes || !block_map || n_blocks <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(ema_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, teacher, student, sizes, block_map, alpha);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ __launch_bounds__(256) void copy_scale_multi_kernel(const float* const* srcs, float* const* dsts, const long long* sizes,
                                                               const int* block_map, float scale) {
    const int ti = block_map[2 * blockIdx.x], start = block_map[2 * blockIdx.x + 1];
    const float* s = srcs[ti];
    float* d = dsts[ti];
    const long long n = sizes[ti];
    const long long end = (long long)start + MT_CHUNK < n ? (long long)start + MT_CHUNK : n;
    constexpr int R = MT_CHUNK / 256;                    // all loads of the chunk before the first store (see sgd_multi_kernel)
    float sv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        sv[r] = i < end ? s[i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = start + threadIdx.x + r * 256;
        if (i < end) d[i] = sv[r] * scale;
    }
}
extern "C" int vs_copy_scale_multi(const float* const* srcs, float* const* dsts, const long long* sizes, const int* block_map,
                                   int n_blocks, float scale, void* stream) {
    if (!srcs || !dsts || !sizes || !block_map || n_blocks <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(copy_scale_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, srcs, dsts, sizes, block_map, scale);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

__global__ void scale_copy_kernel(const float* src, float* dst, long long n, float scale) {     // src may equal dst (in-place scale)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i] * scale;
}
extern "C" int vs_scale_copy(const float* src, float* dst, long long count, float scale, void* stream) {
    if (!src || !dst || count <= 0) return VS_EINVAL;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, count, scale);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- zero fill (statistics arenas): a kernel of this library, not a memset node (common.h: vs_zero_async) -------------------------
__global__ __launch_bounds__(256) void zero_fill_kernel(u32x4* __restrict__ p, long long frags) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < frags; i += (long long)gridDim.x * 256) p[i] = u32x4{0u, 0u, 0u, 0u};
}
extern "C" int vs_zero_fill(void* p, long long bytes, void* stream) {
    if (!p || bytes <= 0) return VS_EINVAL;
    if (((uintptr_t)p & 15) || (bytes & 15)) return VS_EALIGN;
    const long long frags = bytes / 16;
    long long blocks = (frags + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)p, frags);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
