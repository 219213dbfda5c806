// C-ABI launchers of the implicit-GEMM convolutions (forward / backward-data).
#include <stdlib.h>
#include "igemm_dispatch.h"

int g1_dispatch_k3_f32(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s);
int g1_dispatch_k3_bf16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s);
int g1_dispatch_k3_f16(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s);
int g1_k3_fa_supported(const G1Params& p, int ck, int mt);
int k3tw_slab_count(int n, int d, int h, int w);
int k3x_ea_capacity(int n, int c, int m);            // igemm_k3x.hip: the same for the parity mode's k3x_kernel<8, 16, .., EA>
int k3b_ea_capacity(int n, int c, int m);            // igemm_k3_bf16.hip: workgroups of a k3b_kernel<.., EA> launch that are certainly resident together
int k2s2_scatter8_launch(const G1Params& p, int dtype, hipStream_t stream);      // k2s2_scatter8.hip        // igemm_k3_bf16.hip: workgroups (= slabs) of a k3tw_kernel launch
int g1_dispatch_k3_x3(const G1Params& p, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s);

// fp32 parity mode: the 3x3x3 convolutions run on the bf16 matrix cores through exact three-limb operand splitting (igemm_k3x.h); their packed
// weights are VS_F32X3 images (pack.hip).  VS_F32_LIMBS=0 keeps the exact-f32 MFMA kernels (igemm_k3.h) and the plain fp32 images.
extern "C" int vs_conv_k3_f32_limbs(int d, int h, int w, int c_in) {
    const int on = vs_cfg().f32_limbs;
    if (!on) return 0;
    // volumes up to 6^3 with C a multiple of 32 stay on k3s_kernel<float> (igemm_k3s.h: the whole padded sample in LDS, the waves split the taps): a
    // 4x4x16 tile is mostly padding there and the limb kernel would walk C / 16 chunk stages per workgroup (3^3 x 256: 92 us against 26)
    const int small = vs_cfg().k3_small;
    if (small && c_in % 32 == 0 && c_in <= 1024 && (long long)(d + 2) * (h + 2) * (w + 2) <= 512) return 0;
    return 1;
}

static int dispatch_k3(const G1Params& p, int dtype, int ck, int mt, int epi, int tiles, int row_tiles, hipStream_t s) {
    if (dtype == VS_F32 && vs_conv_k3_f32_limbs(p.D, p.H, p.W, p.C)) {
        G1Params q = p;
        const int ckx = vs_k3x_ck(p.C);
        q.nch = p.C / ckx;
        const int rows16 = epi == EPI_SOFTMAX2 ? 16 : p.rb_total * 16;
        // 32-row workgroups (B fragments shared by two row blocks) when that still fills the chip; chunks of 16 channels with a 32-row weight block
        // fill the 160 KB of LDS exactly, so the per-(n,c) tables must be small
        int mtx = 16;
        if (rows16 % 32 == 0 && (long long)tiles * (rows16 / 32) >= 256 && (ckx == 8 || (size_t)2 * p.N * (p.C + p.M) * sizeof(float) <= 3000)) mtx = 32;
        return g1_dispatch_k3_x3(q, ckx, mtx, epi, tiles, rows16 / mtx, s);
    }
    if (dtype == VS_F32) return g1_dispatch_k3_f32(p, ck, mt, epi, tiles, row_tiles, s);
    if (dtype == VS_BF16) return g1_dispatch_k3_bf16(p, ck, mt, epi, tiles, row_tiles, s);
    return g1_dispatch_k3_f16(p, ck, mt, epi, tiles, row_tiles, s);
}

static int pick_mt(int rows16, long long tiles) {
    // largest row tile that still leaves >= VS_MT_MIN_WGS workgroups; 16 when the layer is too small for that.
    // (more rows per wave = more MFMAs per B fragment read from LDS, fewer workgroups)
    // 1024 since round 5 (256 before; same-box A/B with the round's kernels, profiles/r05_ab_mt_min_wgs_*.json: 96^3 2.416 -> 2.406 ms, 160^3 6.03 -> 6.00, fp32 mode
    // 6.169 -> 6.148; 128 -> 2.456): the full-resolution stride-2 / transposed launches run 32-row workgroups, twice as many, instead of 64-row ones
    const int min_wgs = vs_cfg().mt_min_wgs;
    const int cands[3] = {64, 32, 16};
    for (int i = 0; i < 3; ++i) {
        const int mt = cands[i];
        if (rows16 % mt) continue;
        if (tiles * (rows16 / mt) >= min_wgs || mt == 16) return mt;
    }
    return 16;
}

static int check_common(const void* x, const void* w, int n, int d, int h, int w_, int c_in, int dtype) {
    if (!x || !w) return VS_EINVAL;
    if (n <= 0 || d <= 0 || h <= 0 || w_ <= 0) return VS_ESHAPE;
    if (!(c_in == 8 || c_in == 16 || (c_in % 32 == 0 && c_in > 0 && c_in <= 256))) return VS_ESHAPE;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return VS_EALIGN;
    if ((double)n * d * h * w_ * c_in >= 2147483648.0) return VS_ESHAPE;      // kernels index activations with 32-bit element offsets
    return VS_OK;
}

static inline long long g1_ea_capacity() { return 256; }     // g1_kernel<..., epilogue apply>: workgroups of one launch that are certainly resident together (one per CU)

static int gather_impl(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                       void* y, double* y_stats, const void* mask_x, const double* mask_stats, double* sums,
                       int n, int d, int h, int w, int c_in, int m_out, int kind, int dtype, float eps, void* stream,
                       const void* fa_x = nullptr, const double* fa_sums = nullptr, void* fa_dx = nullptr, int* fa_query = nullptr, float* wg_ws = nullptr,
                       unsigned int* ea_sync = nullptr, unsigned int* ea_fault = nullptr, int* ea_query = nullptr, const void* ea_add = nullptr) {
    int rc = check_common(x, w_packed, n, d, h, w, c_in, dtype);
    if (rc) return rc;
    if (!y || m_out <= 0 || m_out % 8 || ((uintptr_t)y & 15)) return VS_EINVAL;
    if (kind != VS_CONV_K3 && kind != VS_CONV_K2S2) return VS_EINVAL;
    if (kind == VS_CONV_K2S2 && ((d | h | w) & 1)) return VS_ESHAPE;
    G1Params p{};
    p.x = x; p.x_stats = x_stats; p.wp = w_packed; p.bias = bias; p.y = y; p.y_stats = y_stats; p.prob = nullptr;
    p.mask_x = mask_x; p.mask_stats = mask_stats; p.sums = sums;
    p.fa_x = fa_x; p.fa_sums = fa_sums; p.fa_dx = fa_dx; p.wg_ws = wg_ws;
    p.ea_sync = ea_sync; p.ea_fault = ea_fault; p.ea_add = ea_add;
    if (sums && (!mask_x || !mask_stats || y_stats)) return VS_EINVAL;
    p.N = n; p.D = d; p.H = h; p.W = w;
    p.C = c_in; p.M = m_out;
    p.rb_total = (m_out + 15) / 16;
    const int ck = c_in < 32 ? c_in : 32;
    p.nch = c_in / ck;
    p.eps = eps;
    p.inv_count_in = 1.0 / ((double)d * h * w);
    long long tiles;
    if (kind == VS_CONV_K3) {
        p.Do = d; p.Ho = h; p.Wo = w;
        p.tyn = (h + 3) / 4; p.txn = (w + 15) / 16;
        p.tiles_per_sample = ((d + 3) / 4) * p.tyn * p.txn;
    } else {
        p.Do = d / 2; p.Ho = h / 2; p.Wo = w / 2;
        p.tyn = p.txn = 0;
        p.tiles_per_sample = vs_ceil_div((long long)p.Do * p.Ho * p.Wo, 256);
        // few workgroups and several 32-channel chunks (the deep stride-2 convs: 8-32 workgroups, one memory round trip per chunk each):
        // 64-voxel column tiles whose four waves split the chunks (g1_kernel, splitw)
        if (ck == 32 && p.nch >= 2 && (long long)p.tiles_per_sample * n * p.rb_total <= 64) {
            p.tiles_per_sample = vs_ceil_div((long long)p.Do * p.Ho * p.Wo, 64);
            p.tyn = 64;
        }
    }
    tiles = (long long)p.tiles_per_sample * n;
    p.inv_count_out = 1.0 / ((double)p.Do * p.Ho * p.Wo);
    const int rows16 = p.rb_total * 16;
    int mt = (kind != VS_CONV_K3 && p.tyn == 64) ? 16 : pick_mt(rows16, tiles);
    // fused apply on 32-channel chunks: every row-block workgroup of a tile repeats the apply pass over the tile's halo, and the kernel runs one
    // workgroup per CU (512 VGPRs) — 32-row tiles halve both the repeats and the workgroup count (24^3 x 32: 144 workgroups, one round)
    if (kind == VS_CONV_K3 && fa_x != nullptr && ck == 32 && mt == 16 && rows16 % 32 == 0) mt = 32;
    const int row_tiles = rows16 / mt;
    if (kind == VS_CONV_K2S2 && (ea_query != nullptr || ea_sync != nullptr)) {
        // g1_kernel's epilogue apply (igemm.h): any storage type; every workgroup of the launch resident — one per CU is certain (up to 256 VGPRs + AGPRs)
        const bool ok = sums != nullptr && mt == 16 && tiles * row_tiles <= g1_ea_capacity();      // the epilogue-apply instantiations: 16-row workgroups
        if (ea_query != nullptr) { *ea_query = ok ? 1 : 0; return VS_OK; }
        if (!ok) return VS_ESHAPE;
        p.ea_items = p.tiles_per_sample * row_tiles;
    }
    if (ea_query != nullptr) {                             // planning only: does k3b_kernel<32, 16, .., EA> take this backward-data launch?  (16-bit storage, 32-channel
        *ea_query = (kind == VS_CONV_K3 && dtype != VS_F32 && ck == 32 && mt == 16 && sums != nullptr && !fa_x &&        // chunks, not a k3s volume, one resident round)
                     !((long long)(d + 2) * (h + 2) * (w + 2) <= 512 && c_in <= 1024) && tiles * row_tiles <= k3b_ea_capacity(n, c_in, m_out)) ? 1 : 0;
        if (kind == VS_CONV_K3 && dtype == VS_F32 && sums != nullptr && !fa_x && vs_conv_k3_f32_limbs(d, h, w, c_in) && vs_k3x_ck(c_in) == 8 && c_in > 8) {
            // parity mode: k3x_kernel<8, 16, .., MULTI, .., EA> — 8-channel chunks, 16-row workgroups (dispatch_k3 takes 32 rows only for launches of >= 256 workgroups)
            const long long rows16 = p.rb_total;
            const bool mt32 = (p.rb_total * 16) % 32 == 0 && tiles * (p.rb_total / 2) >= 256;
            *ea_query = (!mt32 && tiles * rows16 <= k3x_ea_capacity(n, c_in, m_out)) ? 1 : 0;
        }
        return VS_OK;
    }
    if (fa_query != nullptr) {                             // planning only: would a fused-apply launch of this shape find a kernel?
        if (dtype == VS_F32)                               // parity mode: the 8 -> 8 full-resolution layers (k3xt_kernel<..., FA>, igemm_k3x.h)
            *fa_query = kind == VS_CONV_K3 && vs_conv_k3_f32_limbs(d, h, w, c_in) &&
                        ((c_in == 8 && m_out == 8 && n * 8 <= 192 && vs_k3x_toeplitz(8, 8, 27)) ||
                         (c_in == 16 && m_out == 16 && n * 16 <= 192 && vs_k3x_ck(16) == 8));      // k3xt_kernel / k3x_kernel<8, 16, ..., FA> (igemm_k3x.h)
        else
            *fa_query = kind == VS_CONV_K3 ? g1_k3_fa_supported(p, ck, mt) : 0;
        return VS_OK;
    }
    if (kind == VS_CONV_K3) {
        return dispatch_k3(p, dtype, ck, mt, EPI_RAW, (int)tiles, row_tiles, (hipStream_t)stream);
    }
    return g1_dispatch_k2s2(p, dtype, ck, mt, (int)tiles, row_tiles, (hipStream_t)stream);
}

extern "C" int vs_conv_gather_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                  void* y, double* y_stats, int n, int d, int h, int w, int c_in, int m_out,
                                  int kind, int dtype, float eps, void* stream) {
    return gather_impl(x, x_stats, w_packed, bias, y, y_stats, nullptr, nullptr, nullptr, n, d, h, w, c_in, m_out, kind, dtype, eps, stream);
}

extern "C" int vs_conv_gather_bwd_data(const void* x, const void* w_packed, void* y, const void* mask_x,
                                       const double* mask_stats, double* sums, int n, int d, int h, int w, int c_in,
                                       int m_out, int kind, int dtype, float eps, void* stream) {
    if (!mask_x || !mask_stats || !sums) return VS_EINVAL;
    return gather_impl(x, nullptr, w_packed, nullptr, y, nullptr, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, kind, dtype, eps, stream);
}

extern "C" int vs_conv_k3_bwd_data_fused_apply(const void* g, const void* act_x, const double* act_stats, const double* act_sums,
                                               const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums,
                                               void* dx_out, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps,
                                               void* stream) {
    if (!g || !act_x || !act_stats || !act_sums) return VS_EINVAL;
    if ((mask_x == nullptr) != (sums == nullptr) || (mask_x == nullptr) != (mask_stats == nullptr)) return VS_EINVAL;    // all three (lazy conv input) or none
    if (dtype == VS_F32 && !vs_conv_k3_fused_apply_supported(n, d, h, w, c_in, m_out, mask_x != nullptr, dtype)) return VS_ESHAPE;
    if (((uintptr_t)act_x & 15) || (dx_out && ((uintptr_t)dx_out & 15))) return VS_EALIGN;
    return gather_impl(g, act_stats, w_packed, nullptr, y, nullptr, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, VS_CONV_K3, dtype, eps,
                       stream, act_x, act_sums, dx_out);
}

extern "C" int vs_conv_k3_fused_apply_supported(int n, int d, int h, int w, int c_in, int m_out, int lazy_input, int dtype) {
    if (!vs_dtype_ok(dtype)) return 0;
    static const char dummy[16] __attribute__((aligned(16))) = {0};        // planning only: no pointer is dereferenced
    static double dsink[2];
    int ok = 0;
    const int rc = gather_impl(dummy, (const double*)dummy, dummy, nullptr, (void*)dummy, nullptr, lazy_input ? dummy : nullptr,
                               lazy_input ? (const double*)dummy : nullptr, lazy_input ? dsink : nullptr, n, d, h, w, c_in, m_out, VS_CONV_K3,
                               dtype, 1e-5f, nullptr, dummy, (const double*)dummy, nullptr, &ok);
    return rc == VS_OK ? ok : 0;
}

// ---- backward-data whose epilogue applies the InstanceNorm+ReLU backward itself (igemm_k3b.h EA, round 6) ----
extern "C" int vs_conv_k3_bwd_data_applied_supported(int n, int d, int h, int w, int c_in, int m_out, int dtype) {
    const int on = vs_cfg().epilogue_apply;
    if (!on || !vs_dtype_ok(dtype)) return 0;
    static const char dummy[16] __attribute__((aligned(16))) = {0};        // planning only: no pointer is dereferenced
    static double dsink[2];
    int ok = 0;
    const int rc = gather_impl(dummy, nullptr, dummy, nullptr, (void*)dummy, nullptr, dummy, (const double*)dummy, dsink, n, d, h, w, c_in, m_out, VS_CONV_K3, dtype,
                               1e-5f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ok);
    return rc == VS_OK ? ok : 0;
}

extern "C" int vs_conv_k3_bwd_data_applied(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums,
                                           unsigned int* sync, unsigned int* fault, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps, void* stream) {
    if (!mask_x || !mask_stats || !sums || !sync || !fault) return VS_EINVAL;
    if (((uintptr_t)sync & 127) || ((uintptr_t)fault & 3)) return VS_EALIGN;
    if (!vs_conv_k3_bwd_data_applied_supported(n, d, h, w, c_in, m_out, dtype)) return VS_ESHAPE;
    return gather_impl(x, nullptr, w_packed, nullptr, y, nullptr, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, VS_CONV_K3, dtype, eps, stream,
                       nullptr, nullptr, nullptr, nullptr, nullptr, sync, fault);
}

// ---- backward-data with the layer's weight gradient fused (igemm_k3tw.h) ----
extern "C" int vs_conv_k3_bwd_data_wgrad_supported(int n, int d, int h, int w, int c_in, int m_out, int dtype) {
    const int on = vs_cfg().fuse_wgrad;
    if (!on || (dtype != VS_BF16 && dtype != VS_F16)) return 0;
    if (c_in != 8 || m_out != 8 || n <= 0 || n * 8 > 192 || d <= 0 || h <= 0 || w <= 0) return 0;
    if ((double)n * d * h * w * 16 >= 2147483648.0) return 0;
    return vs_k3_toeplitz(8, 8, 27, dtype) ? 1 : 0;     // the weight image is the Toeplitz one of k3t_kernel
}

extern "C" int vs_conv_k3_bwd_data_wgrad_slabs(int n, int d, int h, int w) {
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    return k3tw_slab_count(n, d, h, w);
}

extern "C" int vs_conv_k3_bwd_data_wgrad(const void* g, const void* act_x, const double* act_stats, const double* act_sums, const void* w_packed, void* y,
                                         const void* mask_x, const double* mask_stats, double* sums, float* slabs, int n, int d, int h, int w, int c_in,
                                         int m_out, int dtype, float eps, void* stream) {
    if (!g || !slabs || !mask_x || !mask_stats || !sums) return VS_EINVAL;
    if ((act_x == nullptr) != (act_stats == nullptr) || (act_x == nullptr) != (act_sums == nullptr)) return VS_EINVAL;     // all three (un-applied gradient in) or none
    if (!vs_conv_k3_bwd_data_wgrad_supported(n, d, h, w, c_in, m_out, dtype)) return VS_ESHAPE;
    if (((uintptr_t)slabs & 15) || (act_x && ((uintptr_t)act_x & 15))) return VS_EALIGN;
    return gather_impl(g, act_stats, w_packed, nullptr, y, nullptr, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, VS_CONV_K3, dtype, eps,
                       stream, act_x, act_sums, nullptr, nullptr, slabs);
}

// out_block's backward in one launch: softmax backward while staging, backward-data with the fused IN-backward sums [, weight gradient slabs, bias partials]
extern "C" int vs_conv_k3_softmax2_bwd_data(const float* prob, const float* gprob, const void* gprob_cl, const void* w_packed, void* y, const void* mask_x,
                                            const double* mask_stats, double* sums, float* slabs, double* bias_part, int n, int d, int h, int w, int dtype,
                                            float eps, float drop_p, unsigned long long drop_seed, void* stream) {
    if (!prob || (!gprob && !gprob_cl) || !w_packed || !y || !mask_x || !mask_stats || !sums) return VS_EINVAL;
    if (bias_part && !slabs) return VS_EINVAL;             // the bias partials travel with the weight gradient's slabs
    if (!vs_conv_k3_bwd_data_wgrad_supported(n, d, h, w, 8, 8, dtype)) return VS_ESHAPE;
    if (((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15) || ((uintptr_t)mask_x & 15) || (gprob_cl && ((uintptr_t)gprob_cl & 15)) || (slabs && ((uintptr_t)slabs & 15)) ||
        ((uintptr_t)prob & 3) || ((uintptr_t)gprob & 3)) return VS_EALIGN;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return VS_EINVAL;
    G1Params p{};
    p.wp = w_packed; p.y = y; p.mask_x = mask_x; p.mask_stats = mask_stats; p.sums = sums;
    p.sm_prob = prob; p.sm_gprob = gprob; p.sm_gcl = gprob_cl; p.wg_ws = slabs; p.wg_bias = bias_part;
    p.drop_p = drop_p; p.drop_seed = drop_seed;
    p.N = n; p.D = d; p.H = h; p.W = w; p.Do = d; p.Ho = h; p.Wo = w; p.C = 8; p.M = 8; p.rb_total = 1; p.nch = 1;
    p.eps = eps;
    p.inv_count_in = 1.0 / ((double)d * h * w); p.inv_count_out = p.inv_count_in;
    return dispatch_k3(p, dtype, 8, 16, EPI_RAW, 0, 1, (hipStream_t)stream);
}

static int scatter_impl(const void* x, const double* x_stats, const void* w_packed, const float* bias, void* y,
                        const void* mask_x, const double* mask_stats, double* sums, int n, int d, int h, int w, int c_in,
                        int m_out, int dtype, float eps, void* stream,
                        unsigned int* ea_sync = nullptr, unsigned int* ea_fault = nullptr, int* ea_query = nullptr, const void* ea_add = nullptr) {
    int rc = check_common(x, w_packed, n, d, h, w, c_in, dtype);
    if (rc) return rc;
    if (!y || m_out <= 0 || m_out % 8 || ((uintptr_t)y & 15)) return VS_EINVAL;
    G1Params p{};
    p.x = x; p.x_stats = x_stats; p.wp = w_packed; p.bias = bias; p.y = y; p.y_stats = nullptr; p.prob = nullptr;
    p.mask_x = mask_x; p.mask_stats = mask_stats; p.sums = sums;
    p.inv_count_out = 1.0 / (8.0 * d * h * w);
    p.N = n; p.D = d; p.H = h; p.W = w;
    p.Do = d; p.Ho = h; p.Wo = w;
    p.C = c_in; p.M = m_out;
    p.rb_total = (8 * m_out + 15) / 16;
    const int ck = c_in < 32 ? c_in : 32;
    p.nch = c_in / ck;
    p.eps = eps;
    p.inv_count_in = 1.0 / ((double)d * h * w);
    p.tiles_per_sample = vs_ceil_div((long long)d * h * w, 256);
    const long long tiles = (long long)p.tiles_per_sample * n;
    // the 8 -> 8 backward-data scatter at full resolution (Down1): a streaming kernel instead of an MFMA tile per 256 coarse voxels (k2s2_scatter8.hip)
    const int stream8 = vs_cfg().k2s2_stream;
    const bool ea_asked = ea_query != nullptr || ea_sync != nullptr;
    if (!ea_asked && stream8 && sums && c_in == 8 && m_out == 8 && dtype != VS_F32 && !bias && !x_stats) {
        rc = k2s2_scatter8_launch(p, dtype, (hipStream_t)stream);
        if (rc != VS_ESHAPE) return rc;
    }
    const int rows16 = p.rb_total * 16;
    const int mt = pick_mt(rows16, tiles);
    if (ea_asked) {                                        // g1_kernel's epilogue apply (see gather_impl); not the streaming 8 -> 8 kernel's shapes (full resolution: never resident)
        const bool ok = sums != nullptr && mt == 16 && tiles * (rows16 / mt) <= g1_ea_capacity();
        if (ea_query != nullptr) { *ea_query = ok ? 1 : 0; return VS_OK; }
        if (!ok) return VS_ESHAPE;
        p.ea_sync = ea_sync; p.ea_fault = ea_fault; p.ea_add = ea_add;
        p.ea_items = p.tiles_per_sample * (rows16 / mt);
    }
    return g1_dispatch_pw(p, dtype, ck, mt, (int)tiles, rows16 / mt, (hipStream_t)stream);
}

extern "C" int vs_conv_scatter_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                   void* y, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps,
                                   void* stream) {
    return scatter_impl(x, x_stats, w_packed, bias, y, nullptr, nullptr, nullptr, n, d, h, w, c_in, m_out, dtype, eps, stream);
}

extern "C" int vs_conv_scatter_bwd_data(const void* x, const void* w_packed, void* y, const void* mask_x,
                                        const double* mask_stats, double* sums, int n, int d, int h, int w, int c_in,
                                        int m_out, int dtype, float eps, void* stream) {
    if (!mask_x || !mask_stats || !sums) return VS_EINVAL;
    return scatter_impl(x, nullptr, w_packed, nullptr, y, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, dtype, eps, stream);
}

// ---- the same for the stride-2 kinds (igemm.h g1_kernel's epilogue apply): scatter = 0: backward-data of ConvTranspose3d(k2, s2) (a stride-2 gather);
// scatter = 1: backward-data of Conv3d(k2, s2) (the scatter form).  add (nullable): a second gradient of the same raw tensor, summed in after the apply ----
extern "C" int vs_conv_s2_bwd_data_applied_supported(int n, int d, int h, int w, int c_in, int m_out, int scatter, int dtype) {
    const int on = vs_cfg().epilogue_apply;
    if (!on || !vs_dtype_ok(dtype)) return 0;
    static const char dummy[16] __attribute__((aligned(16))) = {0};        // planning only: no pointer is dereferenced
    static double dsink[2];
    int ok = 0;
    const int rc = scatter ? scatter_impl(dummy, nullptr, dummy, nullptr, (void*)dummy, dummy, (const double*)dummy, dsink, n, d, h, w, c_in, m_out, dtype, 1e-5f, nullptr,
                                          nullptr, nullptr, &ok)
                           : gather_impl(dummy, nullptr, dummy, nullptr, (void*)dummy, nullptr, dummy, (const double*)dummy, dsink, n, d, h, w, c_in, m_out, VS_CONV_K2S2, dtype,
                                         1e-5f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ok);
    return rc == VS_OK ? ok : 0;
}

extern "C" int vs_conv_s2_bwd_data_applied(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums, const void* add,
                                           unsigned int* sync, unsigned int* fault, int n, int d, int h, int w, int c_in, int m_out, int scatter, int dtype, float eps,
                                           void* stream) {
    if (!mask_x || !mask_stats || !sums || !sync || !fault) return VS_EINVAL;
    if (((uintptr_t)sync & 127) || ((uintptr_t)fault & 3) || (add && ((uintptr_t)add & 15))) return VS_EALIGN;
    if (!vs_conv_s2_bwd_data_applied_supported(n, d, h, w, c_in, m_out, scatter, dtype)) return VS_ESHAPE;
    if (scatter) return scatter_impl(x, nullptr, w_packed, nullptr, y, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, dtype, eps, stream, sync, fault, nullptr, add);
    return gather_impl(x, nullptr, w_packed, nullptr, y, nullptr, mask_x, mask_stats, sums, n, d, h, w, c_in, m_out, VS_CONV_K2S2, dtype, eps, stream,
                       nullptr, nullptr, nullptr, nullptr, nullptr, sync, fault, nullptr, add);
}

extern "C" int vs_conv_k3_softmax2_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                       float* prob, int n, int d, int h, int w, int c_in, int dtype, float eps,
                                       void* stream) {
    return vs_conv_k3_softmax2_dropout_fwd(x, x_stats, w_packed, bias, prob, n, d, h, w, c_in, dtype, eps, 0.f, 0ull, stream);
}

extern "C" int vs_conv_k3_softmax2_dropout_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                               float* prob, int n, int d, int h, int w, int c_in, int dtype, float eps,
                                               float drop_p, unsigned long long drop_seed, void* stream) {
    return vs_conv_k3_softmax2_cl_fwd(x, x_stats, w_packed, bias, prob, nullptr, n, d, h, w, c_in, dtype, eps, drop_p, drop_seed, stream);
}

extern "C" int vs_conv_k3_softmax2_cl_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                          float* prob, void* prob_cl, int n, int d, int h, int w, int c_in, int dtype, float eps,
                                          float drop_p, unsigned long long drop_seed, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) return VS_EINVAL;
    if (prob_cl && dtype == VS_F32) return VS_EDTYPE;
    if (prob_cl && ((uintptr_t)prob_cl & 15)) return VS_EALIGN;
    int rc = check_common(x, w_packed, n, d, h, w, c_in, dtype);
    if (rc) return rc;
    if (!prob || c_in != 8) return VS_ESHAPE;
    G1Params p{};
    p.x = x; p.x_stats = x_stats; p.wp = w_packed; p.bias = bias; p.y = prob_cl; p.y_stats = nullptr; p.prob = prob;
    p.mask_x = nullptr; p.mask_stats = nullptr; p.sums = nullptr; p.inv_count_out = 1.0 / ((double)d * h * w);
    p.drop_p = drop_p; p.drop_seed = drop_seed;
    p.N = n; p.D = d; p.H = h; p.W = w;
    p.Do = d; p.Ho = h; p.Wo = w;
    p.C = c_in; p.M = 8;
    p.rb_total = 1;
    p.nch = 1;
    p.eps = eps;
    p.inv_count_in = 1.0 / ((double)d * h * w);
    p.tyn = (h + 3) / 4; p.txn = (w + 15) / 16;
    p.tiles_per_sample = ((d + 3) / 4) * p.tyn * p.txn;
    const long long tiles = (long long)p.tiles_per_sample * n;
    return dispatch_k3(p, dtype, 8, 16, EPI_SOFTMAX2, (int)tiles, 1, (hipStream_t)stream);
}
