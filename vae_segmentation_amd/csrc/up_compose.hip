// Composed Up block (igemm_k4.h): ConvTranspose3d(Cin, Cm, 2, stride 2) -> Conv3d(Cm, Co, 3, padding 1) as one operator.
//   vs_up_compose       Weff[p][o][co][ci] (fp32), the bias table, the MFMA-fragment images of both directions and their tap lists
//   vs_up_conv_fwd      k4t_kernel      vs_up_conv_bwd_data   k4g_kernel
//   vs_up_chain         the parameter-space chain rule: dWeff (+ boundary sums of the output gradient) -> dW3, dW2, db2
#include "igemm_k4.h"

// per axis: neighbour index i in {0,1} of parity p <-> coarse offset o = i - 1 + p; the taps (d, t) with p + d = 2 o + t, d in {-1,0,1}, t in {0,1}
__device__ __host__ static inline int up_axis_pairs(int p, int i, int (&d)[2], int (&t)[2]) {
    const int o = i - 1 + p;
    int n = 0;
    for (int tt = 0; tt < 2; ++tt) {
        const int dd = 2 * o + tt - p;
        if (dd >= -1 && dd <= 1) { d[n] = dd; t[n] = tt; ++n; }
    }
    return n;
}

// Weff[(p * 8 + o8) * Co + co][ci] = sum_{(d,t) in S(p,o)} sum_cm W3[co][cm][d] * W2[ci][cm][t]      (fp32, fixed summation order)
// one workgroup per (p, o8, co): the W3 rows it needs go through LDS, thread = ci
__global__ __launch_bounds__(256) void up_weff_kernel(const float* __restrict__ w2, const float* __restrict__ w3, float* __restrict__ weff,
                                                      int cin, int cm, int co_n) {
    extern __shared__ float s_w3[];                       // [<= 8 taps][cm]
    const int co = blockIdx.x % co_n, po = blockIdx.x / co_n, o8 = po & 7, pp = po >> 3;
    int dz[2], tz[2], dy[2], ty[2], dx[2], tx[2];
    const int nz = up_axis_pairs((pp >> 2) & 1, (o8 >> 2) & 1, dz, tz);
    const int ny = up_axis_pairs((pp >> 1) & 1, (o8 >> 1) & 1, dy, ty);
    const int nx = up_axis_pairs(pp & 1, o8 & 1, dx, tx);
    int t2[8], nt = 0;
    for (int a = 0; a < nz; ++a)
        for (int b = 0; b < ny; ++b)
            for (int c = 0; c < nx; ++c) {
                const int d3 = (dz[a] + 1) * 9 + (dy[b] + 1) * 3 + (dx[c] + 1);
                for (int m = threadIdx.x; m < cm; m += 256) s_w3[nt * cm + m] = w3[((size_t)co * cm + m) * 27 + d3];
                t2[nt++] = tz[a] * 4 + ty[b] * 2 + tx[c];
            }
    __syncthreads();
    for (int ci = threadIdx.x; ci < cin; ci += 256) {
        float acc = 0.f;
        for (int k = 0; k < nt; ++k) {
            const float* wr = w2 + (size_t)ci * cm * 8 + t2[k];
            const float* a3 = s_w3 + k * cm;
            for (int m = 0; m < cm; ++m) acc = fmaf(a3[m], wr[(size_t)m * 8], acc);
        }
        weff[((size_t)po * co_n + co) * cin + ci] = acc;
    }
}

// btab[cls = (cz, cy, cx)][co] = sum over the 3x3x3 taps d that stay inside the fine volume for a voxel of boundary class cls (0: first
// plane, 1: interior, 2: last plane, per axis) of gamma[d][co], gamma[d][co] = sum_cm W3[co][cm][d] * b2[cm].  One block per 8 output channels:
// gamma (coalesced over d) goes through LDS.
__device__ __forceinline__ bool up_tap_leaves(int cls, int d) {
    const int cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3, dz = d / 9 - 1, dy = (d / 3) % 3 - 1, dx = d % 3 - 1;
    return (cz == 0 && dz < 0) || (cz == 2 && dz > 0) || (cy == 0 && dy < 0) || (cy == 2 && dy > 0) || (cx == 0 && dx < 0) || (cx == 2 && dx > 0);
}
__device__ __forceinline__ void up_btab_block(const float* __restrict__ w3, const float* __restrict__ b2, float* __restrict__ btab, int cm, int co_n,
                                              int co0, float* s_gam /* [8][27] */, int tid, int nthr) {
    for (int i = tid; i < 8 * 27; i += nthr) {
        const int c8 = i / 27, d = i - c8 * 27, co = co0 + c8;
        float acc = 0.f;
        if (b2 != nullptr && co < co_n)
            for (int m = 0; m < cm; ++m) acc = fmaf(w3[((size_t)co * cm + m) * 27 + d], b2[m], acc);
        s_gam[i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < 27 * 8; i += nthr) {
        const int cls = i / 8, c8 = i & 7, co = co0 + c8;
        float acc = 0.f;
        for (int d = 0; d < 27; ++d) acc += up_tap_leaves(cls, d) ? 0.f : s_gam[c8 * 27 + d];
        if (co < co_n) btab[cls * co_n + co] = acc;
    }
}
__global__ __launch_bounds__(256) void up_btab_kernel(const float* __restrict__ w3, const float* __restrict__ b2, float* __restrict__ btab, int cm, int co_n) {
    __shared__ float s_gam[8 * 27];
    up_btab_block(w3, b2, btab, cm, co_n, blockIdx.x * 8, s_gam, threadIdx.x, 256);
}

__device__ __forceinline__ float up_weff_at(const float* weff, int pz, int py, int px, int iz, int iy, int ix, int co, int ci, int co_n, int cin) {
    if ((unsigned)iz > 1u || (unsigned)iy > 1u || (unsigned)ix > 1u) return 0.f;
    const int po = ((pz * 4 + py * 2 + px) << 3) | (iz * 4 + iy * 2 + ix);
    return weff[((size_t)po * co_n + co) * cin + ci];
}

// forward image [rb][ch][kg][lane][8] and tap list [rb][NTT]; one thread per 16-byte fragment
struct UpWeffGlobal {
    const float* weff; int cin, co_n;
    __device__ __forceinline__ float at(int pz, int py, int px, int iz, int iy, int ix, int co, int ci) const {
        return up_weff_at(weff, pz, py, px, iz, iy, ix, co, ci, co_n, cin);
    }
};

// ELEM >= 0: compute and store only element ELEM of the fragment (the multi-block small-layer kernel: one thread per element)
template <typename T, typename ACC>
__device__ __forceinline__ void up_pack_fwd_one(const ACC& weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags, const long long f, const int elem = -1) {
    const int CK = cin < 32 ? 16 : 32, NT = CK == 32 ? 8 : 6, nch = cin / CK, nb = co_n >= 16 ? co_n / 16 : 1;
    const int rb_total = 8 * co_n / 16;
    if (f < (long long)rb_total * (CK == 32 ? 8 : 12) && elem <= 0) {          // the tap lists (a few threads)
        const int ntt = CK == 32 ? 8 : 12, rb = (int)(f / ntt), k = (int)(f % ntt);
        int code;
        if (CK == 32) {
            const int pp = rb / nb, pz = (pp >> 2) & 1, py = (pp >> 1) & 1, px = pp & 1;
            const int iz = (k >> 2) & 1, iy = (k >> 1) & 1, ix = k & 1;
            code = (iz + pz) * 9 + (iy + py) * 3 + (ix + px);                        // offset o = i - 1 + p, tap index o + 1
        } else {
            const int pz = (rb >> 1) & 1, py = rb & 1;
            const int iz = k / 6, iy = (k / 3) % 2, oxa = k % 3;                      // (iz, iy, absolute x offset + 1)
            code = (iz + pz) * 9 + (iy + py) * 3 + oxa;
        }
        taps[f] = code;
    }
    if (f >= frags) return;
    long long r = f;
    const int lane = (int)(r % 64); r /= 64;
    const int kg = (int)(r % NT); r /= NT;
    const int ch = (int)(r % nch);
    const int rb = (int)(r / nch);
    const int r16 = lane & 15, g = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float w = 0.f;
        if (elem >= 0 && j != elem) { v[j] = 0.f; continue; }
        if (CK == 32) {
            const int pp = rb / nb, co = (rb % nb) * 16 + r16, ci = ch * 32 + 8 * g + j;
            w = weff.at((pp >> 2) & 1, (pp >> 1) & 1, pp & 1, (kg >> 2) & 1, (kg >> 1) & 1, kg & 1, co, ci);
        } else {
            const int pz = (rb >> 1) & 1, py = rb & 1, px = r16 >> 3, co = r16 & 7;
            const int k = 2 * kg + (g >> 1), iz = k / 6, iy = (k / 3) % 2, oxa = k % 3, ci = (g & 1) * 8 + j;
            w = weff.at(pz, py, px, iz, iy, oxa - px, co, ci);       // i = o + 1 - p with o = oxa - 1
        }
        v[j] = w;
    }
    if (elem >= 0) { float w1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) w1 += v[j];
        ET<T>::st(img + f * 8 + elem, w1); }
    else *(u32x4*)(img + f * 8) = frag_pack(v, (T*)nullptr);
}

// backward-data image [rb (ci / 16)][chunk][j][lane][8] and tap list [chunk][NT]
template <typename T>
__global__ void up_pack_fwd_kernel(const float* __restrict__ weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags) {
    up_pack_fwd_one<T>(UpWeffGlobal{weff, cin, co_n}, img, taps, cin, co_n, frags, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

template <typename T, typename ACC>
__device__ __forceinline__ void up_pack_bwd_one(const ACC& weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags, const long long f, const int elem = -1) {
    const int NT = co_n >= 32 ? 8 : (co_n == 16 ? 12 : 18), nch = 8 * co_n / 32;
    // axis extents of a chunk's delta list: an axis whose parity is fixed by the chunk has 2 deltas, a free one 3
    const int ey = co_n == 8 ? 3 : 2, ex = co_n <= 16 ? 3 : 2;
    auto delta_of = [&](int ch, int j, int& dz, int& dy, int& dx) {
        int pz, py, px;
        if (co_n >= 32) { const int pp = ch / (co_n / 32); pz = (pp >> 2) & 1; py = (pp >> 1) & 1; px = pp & 1; }
        else if (co_n == 16) { pz = (ch >> 1) & 1; py = ch & 1; px = -1; }
        else { pz = ch; py = -1; px = -1; }
        const int jz = j / (ey * ex), jy = (j / ex) % ey, jx = j % ex;
        // fixed parity 0 -> delta in {0, +1}; parity 1 -> {-1, 0}; free -> {-1, 0, +1}
        dz = pz == 0 ? jz : jz - 1;
        dy = py < 0 ? jy - 1 : (py == 0 ? jy : jy - 1);
        dx = px < 0 ? jx - 1 : (px == 0 ? jx : jx - 1);
    };
    if (f < (long long)nch * NT && elem <= 0) {
        int dz, dy, dx;
        delta_of((int)(f / NT), (int)(f % NT), dz, dy, dx);
        taps[f] = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
    }
    if (f >= frags) return;
    long long r = f;
    const int lane = (int)(r % 64); r /= 64;
    const int j = (int)(r % NT); r /= NT;
    const int ch = (int)(r % nch);
    const int rb = (int)(r / nch);
    const int ci = rb * 16 + (lane & 15), g = lane >> 4;
    int dz, dy, dx;
    delta_of(ch, j, dz, dy, dx);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (elem >= 0 && e != elem) { v[e] = 0.f; continue; }
        const int k32 = 8 * g + e;
        int pz, py, px, co;
        if (co_n >= 32) { const int cb = co_n / 32, pp = ch / cb; pz = (pp >> 2) & 1; py = (pp >> 1) & 1; px = pp & 1; co = (ch % cb) * 32 + k32; }
        else if (co_n == 16) { pz = (ch >> 1) & 1; py = ch & 1; px = k32 >> 4; co = k32 & 15; }
        else { pz = ch; py = k32 >> 4; px = (k32 >> 3) & 1; co = k32 & 7; }
        // contribution of fine voxel 2 v' + p to coarse v = v' + o, delta = v' - v = -o; neighbour index i = o + 1 - p
        v[e] = ci < cin ? weff.at(pz, py, px, -dz + 1 - pz, -dy + 1 - py, -dx + 1 - px, co, ci) : 0.f;
    }
    if (elem >= 0) { float w1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) w1 += v[e];
        ET<T>::st(img + f * 8 + elem, w1); }
    else *(u32x4*)(img + f * 8) = frag_pack(v, (T*)nullptr);
}

template <typename T>
__global__ void up_pack_bwd_kernel(const float* __restrict__ weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags) {
    up_pack_bwd_one<T>(UpWeffGlobal{weff, cin, co_n}, img, taps, cin, co_n, frags, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// The whole composition in ONE launch for the small layers (W3 and W2 fit LDS: cin = 16, co = 8 — the large-level Up head, the only one that is
// re-composed every step when its weights train): every block stages W3 / W2 in LDS and packs 256 fragments of the two images, computing each
// Weff element it needs on the spot (<= 8 cm multiply-adds); block 0 also writes the bias table (the tap lists come with the first fragments).
struct UpWeffLds {
    const float* w3; const float* w2; int cin, cm, co_n;
    __device__ __forceinline__ float at(int pz, int py, int px, int iz, int iy, int ix, int co, int ci) const {
        if ((unsigned)iz > 1u || (unsigned)iy > 1u || (unsigned)ix > 1u) return 0.f;
        // (d, t) per axis with p + d = 2 o + t, o = i - 1 + p: enumerated without local arrays (runtime-indexed private arrays live in scratch memory)
        const int oz = iz - 1 + pz, oy = iy - 1 + py, ox = ix - 1 + px;
        float acc = 0.f;
        for (int tz = 0; tz < 2; ++tz) {
            const int dz = 2 * oz + tz - pz;
            if (dz < -1 || dz > 1) continue;
            for (int ty = 0; ty < 2; ++ty) {
                const int dy = 2 * oy + ty - py;
                if (dy < -1 || dy > 1) continue;
                for (int tx = 0; tx < 2; ++tx) {
                    const int dx = 2 * ox + tx - px;
                    if (dx < -1 || dx > 1) continue;
                    const float* a3 = w3 + (size_t)co * cm * 27 + (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
                    const float* a2 = w2 + (size_t)ci * cm * 8 + tz * 4 + ty * 2 + tx;
                    for (int m = 0; m < cm; ++m) acc = fmaf(a3[m * 27], a2[m * 8], acc);
                }
            }
        }
        return acc;
    }
};
template <typename T>
__global__ __launch_bounds__(256) void up_compose_small_kernel(const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
                                                                T* __restrict__ img_f, T* __restrict__ img_b, int* __restrict__ taps_f, int* __restrict__ taps_b,
                                                                float* __restrict__ btab, int cin, int cm, int co_n, long long frags_f, long long frags_b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w3 = (float*)smem;                            // [co][cm][27]
    float* s_w2 = s_w3 + co_n * cm * 27;                   // [cin][cm][8]
    float* s_gam = s_w2 + cin * cm * 8;                    // [8][27]
    const int tid = threadIdx.x;
    {   // both weight tensors into LDS, 8 loads in flight per thread (a load -> LDS store loop waits for every load in turn)
        const int n3 = co_n * cm * 27, n2 = cin * cm * 8;
        for (int i0 = tid; i0 < n3 + n2; i0 += 8 * 256) {
            float r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                r[u] = i < n3 ? w3[i] : (i < n3 + n2 ? w2[i - n3] : 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < n3) s_w3[i] = r[u]; else if (i < n3 + n2) s_w2[i - n3] = r[u];
            }
        }
    }
    __syncthreads();
    const UpWeffLds acc{s_w3, s_w2, cin, cm, co_n};
    const int nbf = (int)((frags_f + 31) / 32);           // 32 fragments (256 elements) per block
    if ((int)blockIdx.x < nbf) up_pack_fwd_one<T>(acc, img_f, taps_f, cin, co_n, frags_f, (long long)blockIdx.x * 32 + (tid >> 3), tid & 7);
    else up_pack_bwd_one<T>(acc, img_b, taps_b, cin, co_n, frags_b, (long long)(blockIdx.x - nbf) * 32 + (tid >> 3), tid & 7);
    if (blockIdx.x == 0)
        for (int c0 = 0; c0 < co_n; c0 += 8) {
            up_btab_block(s_w3, b2, btab, cm, co_n, c0, s_gam, tid, 256);
            __syncthreads();
        }
}

static bool up_shape_ok(int cin, int cm, int co) {
    if (cin != cm) return false;                                       // Up: ConvTranspose3d(C, C)
    if (cin == 16) return co == 8;
    return cin % 32 == 0 && cin <= 256 && co % 16 == 0 && co >= 16 && co <= 128;
}
static void up_sizes(int cin, int co, size_t (&out)[6]) {
    const int CK = cin < 32 ? 16 : 32, NTf = CK == 32 ? 8 : 6, nchf = cin / CK, rbf = 8 * co / 16;
    const int NTb = co >= 32 ? 8 : (co == 16 ? 12 : 18), nchb = 8 * co / 32, rbb = (cin + 15) / 16;
    out[0] = (size_t)64 * co * cin * 4;                                // Weff fp32
    out[1] = (size_t)rbf * nchf * NTf * 64 * 16;                       // forward image
    out[2] = (size_t)rbb * nchb * NTb * 64 * 16;                       // backward-data image
    out[3] = (size_t)rbf * (CK == 32 ? 8 : 12) * 4;                    // forward tap lists
    out[4] = (size_t)nchb * NTb * 4;                                   // backward tap lists
    out[5] = (size_t)27 * co * 4;                                      // bias table
}

extern "C" int vs_up_supported(int cin, int cm, int co, int dtype) {
    return (dtype == VS_BF16 || dtype == VS_F16) && up_shape_ok(cin, cm, co) ? 1 : 0;
}

extern "C" int vs_up_compose_sizes(int cin, int cm, int co, size_t* out6) {
    if (!out6 || !up_shape_ok(cin, cm, co)) return VS_ESHAPE;
    size_t o[6];
    up_sizes(cin, co, o);
    for (int i = 0; i < 6; ++i) out6[i] = o[i];
    return VS_OK;
}

extern "C" int vs_up_compose(const float* w2, const float* b2, const float* w3, float* weff, void* img_fwd, void* img_bwd, int* taps_fwd,
                             int* taps_bwd, float* btab, int cin, int cm, int co, int dtype, void* stream) {
    if (!w2 || !w3 || !weff || !img_fwd || !img_bwd || !taps_fwd || !taps_bwd || !btab) return VS_EINVAL;
    if (!up_shape_ok(cin, cm, co)) return VS_ESHAPE;
    if (dtype != VS_BF16 && dtype != VS_F16) return VS_EDTYPE;
    size_t sz[6];
    up_sizes(cin, co, sz);
    hipStream_t s = (hipStream_t)stream;
    const long long ff0 = (long long)(sz[1] / 16), fb0 = (long long)(sz[2] / 16);
    const size_t small_lds = ((size_t)co * cm * 27 + (size_t)cin * cm * 8 + 8 * 27) * 4;
    if (small_lds <= 48 * 1024 && ff0 >= (long long)(8 * co / 16) * 12 && fb0 >= 64) {      // one launch (the tap lists are written by the first threads of each image)
        const unsigned nb = (unsigned)((ff0 + 31) / 32 + (fb0 + 31) / 32);
        if (dtype == VS_BF16) hipLaunchKernelGGL(up_compose_small_kernel<unsigned short>, dim3(nb), dim3(256), small_lds, s, w2, b2, w3, (unsigned short*)img_fwd, (unsigned short*)img_bwd, taps_fwd, taps_bwd, btab, cin, cm, co, ff0, fb0);
        else hipLaunchKernelGGL(up_compose_small_kernel<vs_half>, dim3(nb), dim3(256), small_lds, s, w2, b2, w3, (vs_half*)img_fwd, (vs_half*)img_bwd, taps_fwd, taps_bwd, btab, cin, cm, co, ff0, fb0);
        VS_CHECK_LAUNCH();
        return VS_OK;
    }
    hipLaunchKernelGGL(up_weff_kernel, dim3(64 * co), dim3(256), (size_t)8 * cm * 4, s, w2, w3, weff, cin, cm, co);
    hipLaunchKernelGGL(up_btab_kernel, dim3((co + 7) / 8), dim3(256), 0, s, w3, b2, btab, cm, co);
    const long long ff = (long long)(sz[1] / 16), fb = (long long)(sz[2] / 16);
    if (dtype == VS_BF16) {
        hipLaunchKernelGGL(up_pack_fwd_kernel<unsigned short>, dim3((unsigned)((ff + 255) / 256)), dim3(256), 0, s, weff, (unsigned short*)img_fwd, taps_fwd, cin, co, ff);
        hipLaunchKernelGGL(up_pack_bwd_kernel<unsigned short>, dim3((unsigned)((fb + 255) / 256)), dim3(256), 0, s, weff, (unsigned short*)img_bwd, taps_bwd, cin, co, fb);
    } else {
        hipLaunchKernelGGL(up_pack_fwd_kernel<vs_half>, dim3((unsigned)((ff + 255) / 256)), dim3(256), 0, s, weff, (vs_half*)img_fwd, taps_fwd, cin, co, ff);
        hipLaunchKernelGGL(up_pack_bwd_kernel<vs_half>, dim3((unsigned)((fb + 255) / 256)), dim3(256), 0, s, weff, (vs_half*)img_bwd, taps_bwd, cin, co, fb);
    }
    VS_CHECK_LAUNCH();
    return VS_OK;
}

static int up_common(G1Params& p, int n, int d, int h, int w, int cin, int co, int dtype, float eps) {
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return VS_ESHAPE;
    if (!up_shape_ok(cin, cin, co)) return VS_ESHAPE;
    if (dtype != VS_BF16 && dtype != VS_F16) return VS_EDTYPE;
    if ((double)n * d * h * w * 8 * co * 2 >= 2147483648.0 || (double)n * d * h * w * cin * 2 >= 2147483648.0) return VS_ESHAPE;    // 32-bit byte offsets
    p.N = n; p.D = d; p.H = h; p.W = w;
    p.Do = d; p.Ho = h; p.Wo = w;
    p.up_co = co;
    p.eps = eps;
    p.tyn = (h + 3) / 4; p.txn = (w + 15) / 16;
    p.tiles_per_sample = ((d + 3) / 4) * p.tyn * p.txn;
    return VS_OK;
}

extern "C" int vs_up_conv_fwd(const void* x, const double* x_stats, const void* img_fwd, const int* taps_fwd, const float* btab, void* y,
                              double* y_stats, int n, int d, int h, int w, int cin, int co, int dtype, float eps, void* stream) {
    if (!x || !img_fwd || !taps_fwd || !btab || !y) return VS_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)img_fwd & 15) || ((uintptr_t)y & 15)) return VS_EALIGN;
    G1Params p{};
    int rc = up_common(p, n, d, h, w, cin, co, dtype, eps);
    if (rc) return rc;
    p.x = x; p.x_stats = x_stats; p.wp = img_fwd; p.up_taps = taps_fwd; p.up_btab = btab; p.y = y; p.y_stats = y_stats;
    p.C = cin; p.M = co;
    p.rb_total = 8 * co / 16;
    p.nch = cin / (cin < 32 ? 16 : 32);
    p.inv_count_in = 1.0 / ((double)d * h * w);
    p.inv_count_out = 1.0 / (8.0 * d * h * w);
    // 4 row blocks per workgroup (the staged tile is shared by them); 2 where that leaves too few workgroups
    const int rb_env = vs_cfg().up_rb;      // tuning knob
    int rb = rb_env ? rb_env : (cin >= 32 ? 2 : 4);
    if (dtype == VS_BF16) return k4t_launch<unsigned short>(p, rb, (hipStream_t)stream);
    return k4t_launch<vs_half>(p, rb, (hipStream_t)stream);
}

extern "C" int vs_up_conv_bwd_data(const void* gy, const void* img_bwd, const int* taps_bwd, void* gx, const void* mask_x, const double* mask_stats,
                                   double* sums, int n, int d, int h, int w, int co, int cin, int dtype, float eps, void* stream) {
    if (!gy || !img_bwd || !taps_bwd || !gx) return VS_EINVAL;
    if ((mask_x == nullptr) != (sums == nullptr) || (mask_x == nullptr) != (mask_stats == nullptr)) return VS_EINVAL;
    if (((uintptr_t)gy & 15) || ((uintptr_t)img_bwd & 15) || ((uintptr_t)gx & 15)) return VS_EALIGN;
    G1Params p{};
    int rc = up_common(p, n, d, h, w, cin, co, dtype, eps);
    if (rc) return rc;
    p.x = gy; p.wp = img_bwd; p.up_taps = taps_bwd; p.y = gx; p.mask_x = mask_x; p.mask_stats = mask_stats; p.sums = sums;
    p.C = 8 * co; p.M = cin;
    p.rb_total = (cin + 15) / 16;
    p.nch = 8 * co / 32;
    p.inv_count_in = 1.0 / (8.0 * d * h * w);
    p.inv_count_out = 1.0 / ((double)d * h * w);
    int mt = 16;
    if (co != 8 && p.rb_total % 2 == 0 && (long long)p.tiles_per_sample * n * (p.rb_total / 2) >= 256) mt = 32;
    if (dtype == VS_BF16) return k4g_launch<unsigned short>(p, mt, (hipStream_t)stream);
    return k4g_launch<vs_half>(p, mt, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// trainable composed block: boundary sums of the output gradient and the parameter-space chain rule
// ---------------------------------------------------------------------------------------------------------------------------------------
// G[cls = (cz, cy, cx)][co] += sum of gy over the fine voxels of boundary class cls (cz: 0 first plane, 1 interior, 2 last plane; the interior class
// (1,1,1) is not summed: nothing needs it).  One workgroup per z-plane (n, z): a face plane is read whole, an interior plane only on its ring
// (rows y = 0 / FH-1 and columns x = 0 / FW-1).  Per-thread fp32 partials over the plane, fixed-order fold in LDS, then one contribution per
// (plane, class, channel) into G, which is a STATISTICS-format buffer double[VS_STAT_SLOTS][27 Co / 2][2] (common.h stat_add: fp64 atomics, or —
// deterministic build — commuting integer atomics on fixed-point limbs): entry e = cls * Co + co is statistic e & 1 of pair e >> 1.
// A face plane is split over blocks of 8 rows (one block per plane made the four face planes the launch's critical path: 36 voxels per thread).
template <typename T>
__global__ __launch_bounds__(256) void up_faces_kernel(const T* __restrict__ gy, double* __restrict__ G, int N, int FD, int FH, int FW, int Co) {
    __shared__ float s_red[4 * 16 * 72];
    const int tid = threadIdx.x, U = Co / 8, cgp = tid % U, lanes = 256 / U, li = tid / U;
    // blocks of one sample: [z = 0: cf chunks of 8 rows][z = FD-1: cf chunks][z = 1 .. FD-2: one block each (the ring only)]
    const int cf = (FH + 7) / 8, per_n = 2 * cf + (FD - 2);
    const int n = blockIdx.x / per_n, rblk = blockIdx.x - n * per_n;
    const int z = rblk < cf ? 0 : (rblk < 2 * cf ? FD - 1 : 1 + (rblk - 2 * cf));
    const int y_lo = rblk < 2 * cf ? (rblk % cf) * 8 : 0, y_hi = rblk < 2 * cf ? (y_lo + 8 < FH ? y_lo + 8 : FH) : FH;
    const int cz = z == 0 ? 0 : (z == FD - 1 ? 2 : 1);
    float acc[3][3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][b][j] = 0.f;
    const T* plane = gy + ((size_t)n * FD + z) * FH * FW * Co + cgp * 8;
    auto add = [&](const u32x4& raw, int y, int x) {
        float f[8];
        frag_unpack(raw, f, (T*)nullptr);
        const int cy = y == 0 ? 0 : (y == FH - 1 ? 2 : 1), cx = x == 0 ? 0 : (x == FW - 1 ? 2 : 1);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const bool hit = a == cy && b == cx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[a][b][j] += hit ? f[j] : 0.f;
            }
    };
    // voxel list of this plane: all of it (a face plane) or its ring (rows y = 0, FH-1 whole; x = 0, FW-1 of the rows between); four loads in
    // flight per thread (one at a time made a face plane a chain of 36 HBM round trips)
    const int ring = 2 * FW + 2 * (FH - 2);
    const int count = cz != 1 ? (y_hi - y_lo) * FW : ring;
    auto coord = [&](int v, int& y, int& x) {
        if (cz != 1) { y = v / FW; x = v - y * FW; y += y_lo; }
        else if (v < FW) { y = 0; x = v; }
        else if (v < 2 * FW) { y = FH - 1; x = v - FW; }
        else { const int q = v - 2 * FW; y = 1 + (q >> 1); x = (q & 1) ? FW - 1 : 0; }
    };
    for (int v0 = li; v0 < count; v0 += 4 * lanes) {
        u32x4 raw[4];
        int yy[4], xx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = v0 + u * lanes;
            yy[u] = 0; xx[u] = 0;
            raw[u] = u32x4{0u, 0u, 0u, 0u};
            if (v < count) {
                coord(v, yy[u], xx[u]);
                raw[u] = *(const u32x4*)(plane + ((size_t)yy[u] * FW + xx[u]) * Co);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add(raw[u], yy[u], xx[u]);        // lanes past the list add zeros
    }
    // fold (fixed order): a butterfly over the lanes of a wave that hold the same channel group (xor offsets 32 .. U), then the four waves through LDS
    const size_t pairs = (size_t)27 * Co / 2;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = acc[a][b][j];
                // constant offsets, uniform predicate: the 72 independent chains interleave (a run-time loop bound serialised them: 432 dependent
                // cross-lane round trips, 23 us)
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const float w = __shfl_xor(v, o, 64); v += o >= U ? w : 0.f; }
                if (lane < U) s_red[(wave * 16 + lane) * 72 + (a * 3 + b) * 8 + j] = v;        // lane = channel group (U <= 16)
            }
    __syncthreads();
    for (int i = tid; i < 72 * U; i += 256) {
        const int g8 = i / 72, r = i - g8 * 72, cls9 = r >> 3, j = r & 7;
        const double tot = ((double)s_red[(0 * 16 + g8) * 72 + r] + (double)s_red[(1 * 16 + g8) * 72 + r]) +
                           ((double)s_red[(2 * 16 + g8) * 72 + r] + (double)s_red[(3 * 16 + g8) * 72 + r]);
        const int cls = cz * 9 + cls9, e = cls * Co + g8 * 8 + j;
        if (cls != 13) stat_add(G, (size_t)(e >> 1), pairs, e & 1, tot);
    }
}

extern "C" int vs_up_faces(const void* gy, double* G, int n, int fd, int fh, int fw, int co, int dtype, void* stream) {
    if (!gy || !G || n <= 0 || fd < 2 || fh < 2 || fw < 2) return VS_EINVAL;
    if (co % 8 || co <= 0 || co > 128 || 256 % (co / 8)) return VS_ESHAPE;
    if (dtype != VS_BF16 && dtype != VS_F16) return VS_EDTYPE;
    if (dtype == VS_BF16) hipLaunchKernelGGL(up_faces_kernel<unsigned short>, dim3(n * (2 * ((fh + 7) / 8) + fd - 2)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gy, G, n, fd, fh, fw, co);
    else hipLaunchKernelGGL(up_faces_kernel<vs_half>, dim3(n * (2 * ((fh + 7) / 8) + fd - 2)), dim3(256), 0, (hipStream_t)stream, (const vs_half*)gy, G, n, fd, fh, fw, co);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// per axis: tap d of the 3x3x3 conv at output parity p reads coarse offset o = floor((p + d) / 2) through transposed-conv tap t = p + d - 2 o
__device__ __forceinline__ void up_axis_ot(int p, int d, int& o, int& t) {
    const int s = p + d;                                  // -1 .. 2
    o = s < 0 ? -1 : (s >> 1);
    t = s - 2 * o;
}

// ONE WAVE per output element of [dw3: co*cm*27][dw2: cin*cm*8][db2: cm]: the lanes split the terms of its sum (independent loads; a thread per
// output made every output a chain of dependent global loads: 150 us for the 5520 outputs of the 16 -> 8 head) and meet in a fixed-order butterfly.
// Hn[d][co] = sum over fine voxels f with f + d INSIDE the volume of gy[f][co] = - sum over the boundary classes for which tap d leaves the volume
// (the volume sum of an InstanceNorm-backward output is zero), from G of vs_up_faces (statistics format).
__global__ __launch_bounds__(256) void up_chain_kernel(const float* __restrict__ dweff, const double* __restrict__ G, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, const float* __restrict__ w3, float* __restrict__ dw2,
                                                       float* __restrict__ db2, float* __restrict__ dw3, int cin, int cm, int co_n) {
    // Hn[d][co] from G (27 statistic loads, all independent): by one lane per class + a butterfly (dw3: one value per wave) or per lane (db2)
    auto g_at = [&](int cls, int co) {
        const int e = cls * co_n + co;
        double v[2];
        stat_load(G, (size_t)(e >> 1), (size_t)27 * co_n / 2, v);
        return v[e & 1];
    };
    const int lane = threadIdx.x & 63;
    const long long n3 = (long long)co_n * cm * 27, n2 = (long long)cin * cm * 8;
    const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e < n3) {
        if (dw3 == nullptr) return;
        const int d = (int)(e % 27), m = (int)((e / 27) % cm), co = (int)(e / (27LL * cm));
        const int dz = d / 9 - 1, dy = (d / 3) % 3 - 1, dx = d % 3 - 1;
        float acc = 0.f;
        for (int t = lane; t < 8 * cin; t += 64) {
            const int pp = t / cin, ci = t - pp * cin;
            int oz, tz, oy, ty, ox, tx;
            up_axis_ot((pp >> 2) & 1, dz, oz, tz); up_axis_ot((pp >> 1) & 1, dy, oy, ty); up_axis_ot(pp & 1, dx, ox, tx);
            const int o27 = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1), t8 = tz * 4 + ty * 2 + tx;
            acc = fmaf(dweff[((size_t)(pp * co_n + co) * cin + ci) * 27 + o27], w2[((size_t)ci * cm + m) * 8 + t8], acc);
        }
        acc = wave_sum(acc);
        if (b2 != nullptr) {
            const double t = (lane < 27 && up_tap_leaves(lane, d)) ? -g_at(lane, co) : 0.0;
            acc = fmaf(b2[m], (float)wave_sum_d(t), acc);
        }
        if (lane == 0) dw3[e] = acc;
    } else if (e < n3 + n2) {
        if (dw2 == nullptr) return;
        const long long r = e - n3;
        const int t8 = (int)(r % 8), m = (int)((r / 8) % cm), ci = (int)(r / (8LL * cm));
        float acc = 0.f;
        for (int t = lane; t < 8 * 27 * co_n; t += 64) {
            const int co = t % co_n, q = t / co_n, d = q % 27, pp = q / 27;
            int oz, tz, oy, ty, ox, tx;
            up_axis_ot((pp >> 2) & 1, d / 9 - 1, oz, tz); up_axis_ot((pp >> 1) & 1, (d / 3) % 3 - 1, oy, ty); up_axis_ot(pp & 1, d % 3 - 1, ox, tx);
            if (tz * 4 + ty * 2 + tx != t8) continue;
            const int o27 = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1);
            acc = fmaf(dweff[((size_t)(pp * co_n + co) * cin + ci) * 27 + o27], w3[((size_t)co * cm + m) * 27 + d], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) dw2[r] = acc;
    } else if (e < n3 + n2 + cm) {
        if (db2 == nullptr) return;
        const int m = (int)(e - n3 - n2);
        float acc = 0.f;
        for (int t = lane; t < 27 * co_n; t += 64) {
            const int d = t / co_n, co = t - d * co_n;
            double h = 0.0;
            for (int cls = 0; cls < 27; ++cls) h -= up_tap_leaves(cls, d) ? g_at(cls, co) : 0.0;
            acc = fmaf(w3[((size_t)co * cm + m) * 27 + d], (float)h, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) db2[m] = acc;
    }
}

extern "C" int vs_up_chain(const float* dweff27, const double* part, const float* w2, const float* b2, const float* w3, float* dw2, float* db2,
                           float* dw3, int cin, int cm, int co, void* stream) {
    if (!dweff27 || !part || !w2 || !w3) return VS_EINVAL;
    if (!up_shape_ok(cin, cm, co)) return VS_ESHAPE;
    const long long total = (long long)co * cm * 27 + (long long)cin * cm * 8 + cm;        // one wave each
    hipLaunchKernelGGL(up_chain_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dweff27, part, w2, b2, w3, dw2, db2, dw3,
                       cin, cm, co);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
