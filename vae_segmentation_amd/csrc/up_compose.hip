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
// plane, 1: interior, 2: last plane, per axis) of sum_cm W3[co][cm][d] * b2[cm]
__global__ void up_btab_kernel(const float* __restrict__ w3, const float* __restrict__ b2, float* __restrict__ btab, int cm, int co_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * co_n) return;
    const int co = i % co_n, cls = i / co_n, cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
    float acc = 0.f;
    if (b2 != nullptr)
        for (int d = 0; d < 27; ++d) {
            const int dz = d / 9 - 1, dy = (d / 3) % 3 - 1, dx = d % 3 - 1;
            if ((cz == 0 && dz < 0) || (cz == 2 && dz > 0) || (cy == 0 && dy < 0) || (cy == 2 && dy > 0) || (cx == 0 && dx < 0) || (cx == 2 && dx > 0)) continue;
            for (int m = 0; m < cm; ++m) acc = fmaf(w3[((size_t)co * cm + m) * 27 + d], b2[m], acc);
        }
    btab[i] = acc;
}

__device__ __forceinline__ float up_weff_at(const float* weff, int pz, int py, int px, int iz, int iy, int ix, int co, int ci, int co_n, int cin) {
    if ((unsigned)iz > 1u || (unsigned)iy > 1u || (unsigned)ix > 1u) return 0.f;
    const int po = ((pz * 4 + py * 2 + px) << 3) | (iz * 4 + iy * 2 + ix);
    return weff[((size_t)po * co_n + co) * cin + ci];
}

// forward image [rb][ch][kg][lane][8] and tap list [rb][NTT]; one thread per 16-byte fragment
template <typename T>
__global__ void up_pack_fwd_kernel(const float* __restrict__ weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CK = cin < 32 ? 16 : 32, NT = CK == 32 ? 8 : 6, nch = cin / CK, nb = co_n >= 16 ? co_n / 16 : 1;
    const int rb_total = 8 * co_n / 16;
    if (f < (long long)rb_total * (CK == 32 ? 8 : 12)) {          // the tap lists (a few threads)
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
        if (CK == 32) {
            const int pp = rb / nb, co = (rb % nb) * 16 + r16, ci = ch * 32 + 8 * g + j;
            w = up_weff_at(weff, (pp >> 2) & 1, (pp >> 1) & 1, pp & 1, (kg >> 2) & 1, (kg >> 1) & 1, kg & 1, co, ci, co_n, cin);
        } else {
            const int pz = (rb >> 1) & 1, py = rb & 1, px = r16 >> 3, co = r16 & 7;
            const int k = 2 * kg + (g >> 1), iz = k / 6, iy = (k / 3) % 2, oxa = k % 3, ci = (g & 1) * 8 + j;
            w = up_weff_at(weff, pz, py, px, iz, iy, oxa - px, co, ci, co_n, cin);       // i = o + 1 - p with o = oxa - 1
        }
        v[j] = w;
    }
    *(u32x4*)(img + f * 8) = frag_pack(v, (T*)nullptr);
}

// backward-data image [rb (ci / 16)][chunk][j][lane][8] and tap list [chunk][NT]
template <typename T>
__global__ void up_pack_bwd_kernel(const float* __restrict__ weff, T* __restrict__ img, int* __restrict__ taps, int cin, int co_n, long long frags) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
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
    if (f < (long long)nch * NT) {
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
        const int k32 = 8 * g + e;
        int pz, py, px, co;
        if (co_n >= 32) { const int cb = co_n / 32, pp = ch / cb; pz = (pp >> 2) & 1; py = (pp >> 1) & 1; px = pp & 1; co = (ch % cb) * 32 + k32; }
        else if (co_n == 16) { pz = (ch >> 1) & 1; py = ch & 1; px = k32 >> 4; co = k32 & 15; }
        else { pz = ch; py = k32 >> 4; px = (k32 >> 3) & 1; co = k32 & 7; }
        // contribution of fine voxel 2 v' + p to coarse v = v' + o, delta = v' - v = -o; neighbour index i = o + 1 - p
        v[e] = ci < cin ? up_weff_at(weff, pz, py, px, -dz + 1 - pz, -dy + 1 - py, -dx + 1 - px, co, ci, co_n, cin) : 0.f;
    }
    *(u32x4*)(img + f * 8) = frag_pack(v, (T*)nullptr);
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
    hipLaunchKernelGGL(up_weff_kernel, dim3(64 * co), dim3(256), (size_t)8 * cm * 4, s, w2, w3, weff, cin, cm, co);
    hipLaunchKernelGGL(up_btab_kernel, dim3((27 * co + 127) / 128), dim3(128), 0, s, w3, b2, btab, cm, co);
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
    static const int rb_env = getenv("VS_UP_RB") ? atoi(getenv("VS_UP_RB")) : 0;      // tuning knob
    int rb = 4;
    if (cin >= 32) rb = rb_env ? rb_env : 2;
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
