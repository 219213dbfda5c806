// C-ABI of the chain kernels (chain.h): the convolutions of one DoubleConv (joint_model.py:35-52) — forward: three 3x3x3 convs, each normalising
// its predecessor's raw output on load; backward: their backward-data convs with the InstanceNorm+ReLU backward between them — in ONE launch.
#include "chain.h"

int chain_dispatch_k3s_bf16(const K3Chain& c, int bwd, hipStream_t s);
int chain_dispatch_k3s_f16(const K3Chain& c, int bwd, hipStream_t s);
int chain_dispatch_k3s_f32(const K3Chain& c, int bwd, hipStream_t s);

static int chain_shape_ok(int n, int d, int h, int w) {
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    return (long long)(d + 2) * (h + 2) * (w + 2) <= 512;                 // k3s_chain_kernel: the whole padded sample chunk in LDS
}

extern "C" int vs_conv_k3_chain_supported(int n, int d, int h, int w, int c_max, int dtype) {
    const int on = vs_cfg().chain;
    if (!on || !vs_dtype_ok(dtype) || !chain_shape_ok(n, d, h, w)) return 0;
    if (c_max <= 0 || c_max % 32 || c_max > 1024) return 0;
    const int v = d * h * w;
    const int cw = ((long long)(d + 2) * (h + 2) * (w + 2) <= 128 && v <= 32) ? 16 * K3S_NCG_SMALL : 16 * K3S_NCG;      // k3s_col_tile()
    const int ctiles = (v + cw - 1) / cw;
    return ctiles * (c_max / 16) <= VS_CHAIN_MAX_ITEMS ? 1 : 0;
}

extern "C" long long vs_conv_k3_chain_sync_bytes(int n) { return n > 0 ? (long long)n * VS_CHAIN_PHASES * 32 * (long long)sizeof(unsigned int) : 0; }

extern "C" int vs_conv_k3_chain(const vs_chain_layer* layers, int n_layers, int backward, const void* add, unsigned int* sync, unsigned int* fault,
                                int n, int d, int h, int w, int dtype, float eps, void* stream) {
    if (!layers || n_layers < 1 || n_layers > VS_CHAIN_MAX_LAYERS || !sync || !fault) return VS_EINVAL;
    if (!vs_dtype_ok(dtype)) return VS_EDTYPE;
    if (!chain_shape_ok(n, d, h, w)) return VS_ESHAPE;
    if (((uintptr_t)sync & 127) || ((uintptr_t)fault & 3) || (add && ((uintptr_t)add & 15))) return VS_EALIGN;
    K3Chain c{};
    c.nl = n_layers; c.sync = sync; c.fault = fault; c.add = add;
    const int es = vs_esize(dtype);
    for (int l = 0; l < n_layers; ++l) {
        const vs_chain_layer& L = layers[l];
        G1Params& p = c.p[l];
        if (!L.x || !L.w_packed || !L.y) return VS_EINVAL;
        if (((uintptr_t)L.x & 15) || ((uintptr_t)L.w_packed & 15) || ((uintptr_t)L.y & 15) || (L.mask_x && ((uintptr_t)L.mask_x & 15))) return VS_EALIGN;
        if (L.c_in <= 0 || L.c_in % 32 || L.m_out <= 0 || L.m_out % 8) return VS_ESHAPE;
        if ((double)n * d * h * w * (L.c_in > L.m_out ? L.c_in : L.m_out) * es >= 2147483648.0) return VS_ESHAPE;
        if (backward) {
            if (L.x_stats || L.y_stats) return VS_EINVAL;
            if ((L.mask_x == nullptr) != (L.sums == nullptr) || (L.mask_x == nullptr) != (L.mask_stats == nullptr)) return VS_EINVAL;
            if (L.apply && !L.sums) return VS_EINVAL;
            if (L.apply) c.apply_mask |= 1 << l;
        } else if (L.mask_x || L.mask_stats || L.sums || L.apply || !L.y_stats) return VS_EINVAL;
        // a layer reads what its predecessor wrote (that is what makes it a chain)
        if (l > 0 && L.x != layers[l - 1].y) return VS_EINVAL;
        if (!backward && l > 0 && L.x_stats != layers[l - 1].y_stats) return VS_EINVAL;
        if (l > 0 && L.c_in != layers[l - 1].m_out) return VS_ESHAPE;
        if (backward && l + 1 < n_layers && !L.apply) return VS_EINVAL;     // only the chain's last gradient may leave un-applied (a stored block input)
        p.x = L.x; p.x_stats = L.x_stats; p.wp = L.w_packed; p.bias = nullptr; p.y = L.y; p.y_stats = L.y_stats;
        p.mask_x = L.mask_x; p.mask_stats = L.mask_stats; p.sums = L.sums;
        p.N = n; p.D = d; p.H = h; p.W = w; p.Do = d; p.Ho = h; p.Wo = w;
        p.C = L.c_in; p.M = L.m_out;
        p.eps = eps;
        p.inv_count_in = p.inv_count_out = 1.0 / ((double)d * h * w);
    }
    if (add && !(backward && (c.apply_mask >> (n_layers - 1)) & 1)) return VS_EINVAL;
    if (dtype == VS_BF16) return chain_dispatch_k3s_bf16(c, backward, (hipStream_t)stream);
    if (dtype == VS_F16) return chain_dispatch_k3s_f16(c, backward, (hipStream_t)stream);
    return chain_dispatch_k3s_f32(c, backward, (hipStream_t)stream);
}
