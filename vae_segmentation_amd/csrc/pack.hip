// Weight packing into MFMA fragment order + layout glue at the planar (NCDHW fp32) boundary.
#include "common.h"

// Packed image: [row-block rb][channel chunk ch][k-group kg][lane][EPL elements]
//   row = rb*16 + (lane & 15);  k within the chunk = kg*KG + (lane >> 4)*EPL + j;  tap = k / CK, c = ch*CK + k % CK
// One thread packs one 16-byte FRAGMENT (EPL consecutive k-side channels of one row / tap): EPL independent gathers, one vector
// store (one element per thread meant 2-byte stores and eight times the threads: 27 us for the step's 136 images).
template <typename T>
__device__ __forceinline__ void pack_one(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1, int ntaps,
                                         int c_pad, int form, long long total, long long frag) {
    constexpr int EPL = ET<T>::EPL, KG = ET<T>::KG;
    const long long i = frag * EPL;                      // first element of the fragment
    if (i >= total) return;
    float v[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) v[j] = 0.f;
    if (sizeof(T) == 2 && form != VS_PACK_SCATTER_D1 && vs_k3_toeplitz(form == VS_PACK_ROWS_D0 ? d0 : d1, c_pad, ntaps, VS_BF16)) {
        // Toeplitz image of the 8-channel 3x3x3 layers (common.h vs_k3_toeplitz, igemm_k3t.h): [kg = tz*3+ty][lane][ci]
        const int lane = (int)(frag & 63), kg = (int)(frag >> 6);
        const int row = lane & 15, dx2 = row >> 3, co = row & 7, tx = (lane >> 4) - dx2;
        if (tx >= 0 && tx <= 2) {
            const int tap = kg * 3 + tx;
#pragma unroll
            for (int ci = 0; ci < EPL; ++ci) {
                if (form == VS_PACK_ROWS_D0) { if (co < d0 && ci < d1) v[ci] = src[((size_t)co * d1 + ci) * 27 + tap]; }
                else { if (co < d1 && ci < d0) v[ci] = src[((size_t)ci * d1 + co) * 27 + (26 - tap)]; }
            }
        }
    } else if (sizeof(T) == 4 && form != VS_PACK_SCATTER_D1 && vs_k3_toeplitz_f32(form == VS_PACK_ROWS_D0 ? d0 : d1, c_pad, ntaps, VS_F32)) {
        // y-Toeplitz image of the same layers in fp32 (common.h vs_k3_toeplitz_f32, igemm_k3.h TY): [kg][lane][4]
        const int lane = (int)(frag & 63), kg = (int)(frag >> 6);
        const int row = lane & 15, dy2 = row >> 3, co = row & 7, g = lane >> 4;
        const int t = kg * 2 + (g >> 1), ci0 = (g & 1) * 4;
        const int dz = t / 12, wy = (t / 3) % 4, dx = t % 3, dy = wy - dy2;
        if (dy >= 0 && dy <= 2) {
            const int tap = dz * 9 + dy * 3 + dx;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int ci = ci0 + j;
                if (form == VS_PACK_ROWS_D0) { if (co < d0 && ci < d1) v[j] = src[((size_t)co * d1 + ci) * 27 + tap]; }
                else { if (co < d1 && ci < d0) v[j] = src[((size_t)ci * d1 + co) * 27 + (26 - tap)]; }
            }
        }
    } else {
        const int CK = c_pad < 32 ? c_pad : 32;
        const int nch = c_pad / CK;
        const int gemm_taps = form == VS_PACK_SCATTER_D1 ? 1 : ntaps;
        const int nkg = (gemm_taps * CK + KG - 1) / KG;
        // 32-bit index arithmetic (an image has fewer than 2^31 fragments: its block count is an int)
        const unsigned int fr = (unsigned int)frag;
        const int lane = (int)(fr & 63u);
        unsigned int r = fr >> 6;
        const int kg = (int)(r % (unsigned int)nkg); r /= (unsigned int)nkg;
        const int rb = (int)(r / (unsigned int)nch);
        const int ch = (int)(r - (unsigned int)rb * (unsigned int)nch);
        const int row = rb * 16 + (lane & 15);
        const int kk0 = kg * KG + (lane >> 4) * EPL;     // EPL divides CK: the fragment stays inside one tap
        const int tap = kk0 / CK;
        const int c0 = ch * CK + kk0 % CK;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int c = c0 + j;
            if (form == VS_PACK_ROWS_D0) {
                if (row < d0 && tap < ntaps && c < d1) v[j] = src[((size_t)row * d1 + c) * ntaps + tap];
            } else if (form == VS_PACK_ROWS_D1_FLIP) {
                if (row < d1 && tap < ntaps && c < d0) v[j] = src[((size_t)c * d1 + row) * ntaps + (ntaps - 1 - tap)];
            } else {  // VS_PACK_SCATTER_D1: rows (t, m = d1 index), k = c = d0 index
                const int t = row / d1, m = row - t * d1;
                if (t < ntaps && tap < 1 && c < d0) v[j] = src[((size_t)c * d1 + m) * ntaps + t];
            }
        }
    }
    *(u32x4*)(dst + i) = frag_pack(v, (T*)nullptr);
}

// VS_F32X3 image of a 3x3x3 weight for k3x_kernel (igemm_k3x.h): [row block][chunk of CK = min(c_pad, 16) channels][k-group of 32][limb 0..2][lane][8 bf16]
//   row = rb*16 + (lane & 15);  k within the chunk = kg*32 + (lane >> 4)*8 + j;  tap = k / CK, c = ch*CK + k % CK
// limb 0 = w rounded to bf16, limb 1 = w - limb0 rounded to bf16, limb 2 = the rest (8 + 8 + 8 significant bits).  One thread per 16-byte fragment.
__device__ __forceinline__ unsigned short limb_of(float v, int limb) {       // round-to-nearest limbs, as vs_limb_split4 (common.h) forms them for the activations
    const unsigned short b0 = f2bf(v);
    const float r1 = v - bf2f(b0);
    const unsigned short b1 = f2bf(r1);
    return limb == 0 ? b0 : (limb == 1 ? b1 : f2bf(r1 - bf2f(b1)));
}

// CK < 0: the Toeplitz limb image of the 8-channel layers (common.h vs_k3x_toeplitz, igemm_k3x.h k3xt_kernel): [kg = tz*3+ty][limb][lane][ci]
__device__ __forceinline__ void pack_one_limbs(const float* __restrict__ src, unsigned short* __restrict__ dst, int d0, int d1, int ntaps,
                                               int c_pad, int form, long long total, long long frag, int CK) {
    const long long i = frag * 8;
    if (i >= total) return;
    if (CK < 0) {
        const int lane = (int)(frag & 63), limb = (int)((frag >> 6) % 3), kg = (int)(frag / 192);
        const int row = lane & 15, dx2 = row >> 3, co = row & 7, tx = (lane >> 4) - dx2;
        unsigned short o[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            float v = 0.f;
            if (tx >= 0 && tx <= 2) {
                const int tap = kg * 3 + tx;
                if (form == VS_PACK_ROWS_D0) { if (co < d0 && ci < d1) v = src[((size_t)co * d1 + ci) * 27 + tap]; }
                else { if (co < d1 && ci < d0) v = src[((size_t)ci * d1 + co) * 27 + (26 - tap)]; }
            }
            o[ci] = limb_of(v, limb);
        }
        u32x4 pk;
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[q] = (unsigned int)o[2 * q] | ((unsigned int)o[2 * q + 1] << 16);
        *(u32x4*)(dst + i) = pk;
        return;
    }
    const int nch = c_pad / CK;
    const int nkg = (ntaps * CK + 31) / 32;
    const unsigned int fr = (unsigned int)frag;         // 32-bit index arithmetic, as pack_one
    const int lane = (int)(fr & 63u);
    unsigned int r = fr >> 6;
    const int limb = (int)(r % 3u); r /= 3u;
    const int kg = (int)(r % (unsigned int)nkg); r /= (unsigned int)nkg;
    const int rb = (int)(r / (unsigned int)nch);
    const int ch = (int)(r - (unsigned int)rb * (unsigned int)nch);
    const int row = rb * 16 + (lane & 15);
    const int kk0 = kg * 32 + (lane >> 4) * 8;           // 8 divides CK: the fragment stays inside one tap
    const int tap = kk0 / CK;
    const int c0 = ch * CK + kk0 % CK;
    unsigned short o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        float v = 0.f;
        if (form == VS_PACK_ROWS_D0) { if (row < d0 && tap < ntaps && c < d1) v = src[((size_t)row * d1 + c) * ntaps + tap]; }
        else { if (row < d1 && tap < ntaps && c < d0) v = src[((size_t)c * d1 + row) * ntaps + (ntaps - 1 - tap)]; }
        o[j] = limb_of(v, limb);
    }
    u32x4 pk;
#pragma unroll
    for (int q = 0; q < 4; ++q) pk[q] = (unsigned int)o[2 * q] | ((unsigned int)o[2 * q + 1] << 16);
    *(u32x4*)(dst + i) = pk;
}
__global__ void pack_weight_limbs_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int d0, int d1, int ntaps, int c_pad, int form,
                                         long long total, int ck) {
    pack_one_limbs(src, dst, d0, d1, ntaps, c_pad, form, total, (long long)blockIdx.x * blockDim.x + threadIdx.x, ck);
}

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1, int ntaps,
                                   int c_pad, int form, long long total) {
    pack_one<T>(src, dst, d0, d1, ntaps, c_pad, form, total, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// descs[].first_block counts 256-thread blocks of FRAGMENTS (vs_pack_desc::total / EPL of them per image)
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const vs_pack_desc* __restrict__ descs, int n_desc, int total_blocks, int k3x_ck) {
    // XCD-aware block order: a fragment gathers 8 floats that lie 27 taps (108 bytes) apart, so the 27 k-groups of one (row block,
    // channel chunk) — seven consecutive blocks — read the same cache lines.  Consecutive hardware block ids go to different XCDs
    // (8, each with its own L2), which made every XCD fetch those lines for itself: 103 MB of HBM traffic for 9 MB of weights.
    // XCD x takes the contiguous run [x * chunk, (x + 1) * chunk) of the logical block list instead (grid = 8 * chunk).
    const int chunk = (int)gridDim.x >> 3;
    const int lb = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    if (lb >= total_blocks) return;
    // binary search: last descriptor whose first_block <= lb.  The first_block column goes through LDS first (one coalesced round trip instead of log2(n_desc)
    // dependent loads at the head of every workgroup; the launch — 18 us for 136 images, bound by its 8 strided 4-byte gathers per fragment — did not move).
    __shared__ int s_fb[1024];
    const bool in_lds = n_desc <= 1024;                  // uniform
    if (in_lds) {
        for (int i = threadIdx.x; i < n_desc; i += 256) s_fb[i] = descs[i].first_block;
        __syncthreads();
    }
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((in_lds ? s_fb[mid] : descs[mid].first_block) <= lb) lo = mid; else hi = mid - 1;
    }
    const vs_pack_desc d = descs[lo];
    const long long i = (long long)(lb - d.first_block) * 256 + threadIdx.x;
    if (d.dtype == VS_F32X3) {
        const int rows = d.form == VS_PACK_ROWS_D0 ? d.d0 : d.d1;
        const bool toep = k3x_ck < 0 ? false : (d.total == 9ll * 3 * 64 * 8 && d.c_pad == 8 && rows <= 8);      // the Toeplitz image has its own size: 13,824 elements
        pack_one_limbs(d.src, (unsigned short*)d.dst, d.d0, d.d1, d.ntaps, d.c_pad, d.form, d.total, i, toep ? -1 : (d.c_pad < k3x_ck ? d.c_pad : k3x_ck));
    }
    else if (d.dtype == VS_F32) pack_one<float>(d.src, (float*)d.dst, d.d0, d.d1, d.ntaps, d.c_pad, d.form, d.total, i);
    else if (d.dtype == VS_BF16) pack_one<unsigned short>(d.src, (unsigned short*)d.dst, d.d0, d.d1, d.ntaps, d.c_pad, d.form, d.total, i);
    else pack_one<vs_half>(d.src, (vs_half*)d.dst, d.d0, d.d1, d.ntaps, d.c_pad, d.form, d.total, i);
}

extern "C" int vs_pack_weight_multi(const vs_pack_desc* descs, int n_desc, int total_blocks, void* stream) {
    if (!descs || n_desc <= 0 || total_blocks <= 0) return VS_EINVAL;
    hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(8 * ((total_blocks + 7) / 8)), dim3(256), 0, (hipStream_t)stream, descs, n_desc, total_blocks, vs_k3x_ck(16));
    VS_CHECK_LAUNCH();
    return VS_OK;
}

static long long packed_elems(int rows, int c_pad, int gemm_taps, int dtype) {
    if (dtype == VS_F32X3) {                             // bf16 elements of the three-limb image (3x3x3 weights only)
        if (vs_k3x_toeplitz(rows, c_pad, gemm_taps)) return 9ll * 3 * 64 * 8;
        const int CK = vs_k3x_ck(c_pad);
        return (long long)((rows + 15) / 16) * (c_pad / CK) * ((gemm_taps * CK + 31) / 32) * 3 * 64 * 8;
    }
    if (vs_k3_toeplitz(rows, c_pad, gemm_taps, dtype)) return 9 * 64 * 8;
    if (vs_k3_toeplitz_f32(rows, c_pad, gemm_taps, dtype)) return 18 * 64 * 4;
    const int EPL = dtype == VS_F32 ? 4 : 8, KG = 4 * EPL;
    const int CK = c_pad < 32 ? c_pad : 32;
    const int nch = c_pad / CK;
    const int nkg = (gemm_taps * CK + KG - 1) / KG;
    const int rbt = (rows + 15) / 16;
    return (long long)rbt * nch * nkg * 64 * EPL;
}

extern "C" size_t vs_packed_weight_bytes(int rows, int c_pad, int ntaps, int dtype) {
    return (size_t)packed_elems(rows, c_pad, ntaps, dtype) * (dtype == VS_F32 ? 4 : 2);
}

extern "C" int vs_pack_weight(const float* src, void* dst, int d0, int d1, int ntaps, int c_pad, int form, int dtype,
                              void* stream) {
    if (!src || !dst || d0 <= 0 || d1 <= 0) return VS_EINVAL;
    if (!(c_pad == 8 || c_pad == 16 || (c_pad % 32 == 0 && c_pad > 0))) return VS_ESHAPE;
    if (!vs_dtype_ok(dtype) && dtype != VS_F32X3) return VS_EDTYPE;
    if (dtype == VS_F32X3 && (form == VS_PACK_SCATTER_D1 || ntaps != 27)) return VS_ESHAPE;      // three-limb images exist for the 3x3x3 kernels only
    int rows, kc, gemm_taps;
    if (form == VS_PACK_ROWS_D0) { rows = d0; kc = d1; gemm_taps = ntaps; }
    else if (form == VS_PACK_ROWS_D1_FLIP) { rows = d1; kc = d0; gemm_taps = ntaps; }
    else if (form == VS_PACK_SCATTER_D1) { rows = ntaps * d1; kc = d0; gemm_taps = 1; }
    else return VS_EINVAL;
    if (kc > c_pad) return VS_ESHAPE;
    if (!(ntaps == 27 || ntaps == 8)) return VS_ESHAPE;
    const long long total = packed_elems(rows, c_pad, gemm_taps, dtype);
    const int blocks = vs_ceil_div(total / (dtype == VS_F32 ? 4 : 8), 256);      // one thread per 16-byte fragment
    if (dtype == VS_F32X3) {
        hipLaunchKernelGGL(pack_weight_limbs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, d0, d1, ntaps, c_pad, form, total,
                           vs_k3x_toeplitz(rows, c_pad, gemm_taps) ? -1 : vs_k3x_ck(c_pad));
        VS_CHECK_LAUNCH();
        return VS_OK;
    }
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, d0, d1, ntaps, c_pad, form, total);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- planar fp32 [N][Cs][V]  <->  channels-last [N][V][Cp] ---------------------------------------
template <typename T>
__global__ void pack_planar_kernel(const float* __restrict__ src, T* __restrict__ dst, long long voxels, int c_src,
                                   int c_pad, long long total_vox) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, v)
    if (i >= total_vox) return;
    const long long n = i / voxels, v = i - n * voxels;
    T* o = dst + i * c_pad;
    for (int c0 = 0; c0 < c_pad; c0 += ET<T>::EPL) {
        float f[ET<T>::EPL];
#pragma unroll
        for (int j = 0; j < ET<T>::EPL; ++j) {
            const int c = c0 + j;
            f[j] = c < c_src ? src[(n * c_src + c) * voxels + v] : 0.f;
        }
        *(u32x4*)(o + c0) = frag_pack(f, (T*)nullptr);
    }
}

template <typename T>
__global__ void unpack_planar_kernel(const T* __restrict__ src, float* __restrict__ dst, long long voxels, int c_dst,
                                     int c_pad, long long total_vox) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_vox) return;
    const long long n = i / voxels, v = i - n * voxels;
    for (int c0 = 0; c0 < c_dst; c0 += ET<T>::EPL) {
        float f[ET<T>::EPL];
        frag_unpack(*(const u32x4*)(src + i * c_pad + c0), f, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < ET<T>::EPL; ++j)
            if (c0 + j < c_dst) dst[(n * c_dst + c0 + j) * voxels + v] = f[j];
    }
}

extern "C" int vs_pack_planar(const float* src, void* dst, int n, long long voxels, int c_src, int c_pad, int dtype,
                              void* stream) {
    if (!src || !dst || n <= 0 || voxels <= 0 || c_src <= 0 || c_src > c_pad || c_pad % 8) return VS_EINVAL;
    const long long tv = (long long)n * voxels;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(pack_planar_kernel<T>, dim3(vs_ceil_div(tv, 256)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, voxels, c_src, c_pad, tv);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}

extern "C" int vs_unpack_planar(const void* src, float* dst, int n, long long voxels, int c_dst, int c_pad, int dtype,
                                void* stream) {
    if (!src || !dst || n <= 0 || voxels <= 0 || c_dst <= 0 || c_dst > c_pad || c_pad % 8) return VS_EINVAL;
    const long long tv = (long long)n * voxels;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    dispatch_t(dtype, [&](auto* tag) {
        using T = TAG_T(tag);
        hipLaunchKernelGGL(unpack_planar_kernel<T>, dim3(vs_ceil_div(tv, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, voxels, c_dst, c_pad, tv);
    });
    VS_CHECK_LAUNCH();
    return VS_OK;
}
