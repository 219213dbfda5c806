/* libvaeseg — C ABI of the MI355X (gfx950) kernels behind the joint_model.py surface.
 *
 * The reference (yyNoBug/VAE_segmentation) has no FFI of its own: its hot path is the stock ATen ops
 * that joint_model.py / utils/evaluation.py invoke.  Each entry point below replaces one such op (or a
 * fused group of them); the reference call site it stands in for is cited as file:line under
 * /root/reference.  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); nothing is allocated, freed or
 *     synchronised inside; every launch goes to `stream` (a hipStream_t passed as void*).
 *   - activations are channels-last: [N][D][H][W][C], C a multiple of 8, element type `dtype`
 *     (VS_F32, VS_BF16 or VS_F16; accumulation is always fp32).  "planar" tensors are the reference's NCDHW fp32: [N][C][D*H*W].
 *   - a *lazy* activation is a raw conv output plus `stats`: double[VS_STAT_SLOTS][N][C][2] = VS_STAT_SLOTS partial copies of
 *     (sum, sum of squares) over the D*H*W voxels of each (n,c): a producing workgroup accumulates (fp64 atomics) into copy
 *     (workgroup id mod VS_STAT_SLOTS), consumers add the copies — same-address atomics retire ~20 ns apart, and hundreds of
 *     workgroups finish together at the full-resolution levels.  The IN-backward `sums` buffers have the same shape.  Passing `stats` to a
 *     consumer makes it read relu((x-mean)*rstd) — InstanceNorm3d(affine=False, eps) + ReLU
 *     (joint_model.py:11,41-48,107-108) — without that tensor ever being materialised.  stats == NULL
 *     means "use x as is".
 *   - return value: 0 on success, negative VS_E* on a rejected call, positive = hipError_t.
 *   - the library is re-entrant: no mutable globals.
 */
#ifndef VAESEG_H
#define VAESEG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_VERSION 500

enum { VS_F32 = 0, VS_BF16 = 1, VS_F16 = 2, VS_F32X3 = 3 /* PACK-ONLY: three-bf16-limb image of an fp32 3x3x3 weight, see vs_conv_k3_f32_limbs */ };   /* storage type of activations: fp32 (parity mode), bf16, IEEE fp16 (needs loss scaling, see vs_loss_scale_*) */
#ifndef VS_STAT_SLOTS
#define VS_STAT_SLOTS 4        /* measured on one MI355X, 96^3 step: 1 copy 2.908 ms, 2: 2.820, 4: 2.797, 8: 3.021 (consumers read every copy) */
#endif
enum { VS_OK = 0, VS_EINVAL = -1, VS_ESHAPE = -2, VS_EDTYPE = -3, VS_EWORKSPACE = -4, VS_EALIGN = -5 };

/* geometry of an implicit-GEMM convolution */
enum {
    VS_CONV_K3 = 0,     /* 3x3x3, stride 1, pad 1          nn.Conv3d(...,3,padding=1)        joint_model.py:40,43,46,106,224,366 */
    VS_CONV_K2S2 = 1,   /* 2x2x2, stride 2, pad 0          nn.Conv3d(C,C,2,stride=2)         joint_model.py:130 */
    VS_CONV_T2S2 = 2,   /* 2x2x2 transposed, stride 2      nn.ConvTranspose3d(C,C,2,stride=2) joint_model.py:118 */
    VS_WGRAD_SLABS = 16, /* vs_wgrad_desc only: p = the slabs vs_conv_k3_bwd_data_wgrad wrote, n = their count (vs_conv_k3_bwd_data_wgrad_slabs); m_ch = c_ch = 8,
                            m_real / c_real / dw as for the layer; the descriptor only takes part in the grouped reduction and must be the only one for its dw */
    VS_CONV_UP = 3      /* vs_wgrad_desc only: the composed Up block (vs_up_*): p = the FINE output gradient (n,2dp,2hp,2wp,Co) read space-to-depth
                           (m_ch = m_real = 8 Co view channels (parity, co); reserved_ = Co), q = the coarse input, dw = dWeff [8 Co][c_real][27] */
};

/* how a weight tensor src[d0][d1][ntaps] (fp32, the reference's parameter layout) is packed into MFMA
 * fragment order for a GEMM whose rows are output channels m and whose k runs over (tap, channel c) */
enum {
    VS_PACK_ROWS_D0 = 0,       /* m = d0, c = d1                 : Conv3d forward; ConvTranspose3d backward-data        */
    VS_PACK_ROWS_D1_FLIP = 1,  /* m = d1, c = d0, taps mirrored  : Conv3d(k3,p1) backward-data                          */
    VS_PACK_SCATTER_D1 = 2     /* rows (tap, m = d1), k = c = d0 : ConvTranspose3d forward; Conv3d(k2,s2) backward-data */
};

int vs_version(void);
int vs_stat_slots(void);        /* VS_STAT_SLOTS the library was built with (callers size statistics buffers by it) */
int vs_stat_interleaved(void);  /* 0: double[slot][N][C][2]   1: double[N][C][slot][2] */
const char* vs_strerror(int code);
/* 1 for the deterministic build of the library (libvaeseg_det.so, -DVS_DET_BUILD=1: parity runs), 0 for the throughput build.  In the
 * deterministic build every per-(n,c) statistic — the conv epilogues' (sum, sumsq), the fused InstanceNorm-backward sums, the vs_instnorm_*
 * reductions — is accumulated with commuting integer atomics on four fixed-point limbs held in the same double[VS_STAT_SLOTS][N][C][2]
 * buffer (csrc/common.h), and the few remaining floating-point atomics (vs_dice_fwd, vs_bce_fwd, vs_bias_grad) run as one block: two runs of a
 * step on the same inputs agree bit for bit.  A statistics buffer must be produced and consumed by the same build. */
int vs_get_deterministic(void);

/* ---- composed Up block (csrc/igemm_k4.h): ConvTranspose3d(C, C, 2, stride 2) -> Conv3d(C, Co, 3, padding 1) as ONE operator -----------------
 * Replaces, for 16-bit storage, the pair vs_conv_scatter_fwd + vs_conv_gather_fwd(K3) behind `Up` (/root/reference/joint_model.py:116-120: the
 * Sequential has nothing between the two convolutions) and, backward, vs_conv_gather_bwd_data(K3) + vs_conv_gather_bwd_data(K2S2).
 * w2 = ConvTranspose3d.weight [cin][cm][2][2][2], b2 = its bias [cm] (nullable), w3 = Conv3d.weight [co][cm][3][3][3] (bias dead: InstanceNorm follows).
 * vs_up_compose_sizes -> bytes of {Weff fp32 scratch, forward image, backward-data image, forward tap lists, backward tap lists, bias table}. */
int vs_up_supported(int cin, int cm, int co, int dtype);
int vs_up_compose_sizes(int cin, int cm, int co, size_t* out6);
int vs_up_compose(const float* w2, const float* b2, const float* w3, float* weff, void* img_fwd, void* img_bwd, int* taps_fwd, int* taps_bwd,
                  float* btab, int cin, int cm, int co, int dtype, void* stream);
/* x: coarse (n,d,h,w,cin) [lazy with x_stats]; y: fine raw conv output (n,2d,2h,2w,co) + its statistics */
int vs_up_conv_fwd(const void* x, const double* x_stats, const void* img_fwd, const int* taps_fwd, const float* btab, void* y, double* y_stats,
                   int n, int d, int h, int w, int cin, int co, int dtype, float eps, void* stream);
/* Weight gradients of the composed block: vs_conv_wgrad_multi with a VS_CONV_UP descriptor gives dWeff27[(parity, co)][ci][27 coarse offsets];
 * vs_up_faces sums the output gradient over the 26 boundary classes of the fine volume (the transposed conv's bias acts only through the zero
 * padding of the 3x3x3 conv: its gradient is a boundary quantity, and the sum over the whole volume of an InstanceNorm-backward output is 0)
 * into G, a statistics-format buffer double[VS_STAT_SLOTS][27 co / 2][2] (ACCUMULATED: caller zeroes; entry cls * co + c = statistic (e & 1) of
 * pair e >> 1; a second use of the weights in the same pass simply adds); vs_up_chain applies the parameter-space chain rule:
 * dw3 [co][cm][27], dw2 [cin][cm][8], db2 [cm] (any of them nullable). */
int vs_up_faces(const void* gy, double* G, int n, int fd, int fh, int fw, int co, int dtype, void* stream);
int vs_up_chain(const float* dweff27, const double* G, const float* w2, const float* b2, const float* w3, float* dw2, float* db2, float* dw3,
                int cin, int cm, int co, void* stream);
/* gy: gradient of y (fine, applied); gx: gradient of the coarse input (n,d,h,w,cin); mask_* / sums: fused InstanceNorm-backward sums of a lazy input (all or none) */
int vs_up_conv_bwd_data(const void* gy, const void* img_bwd, const int* taps_bwd, void* gx, const void* mask_x, const double* mask_stats,
                        double* sums, int n, int d, int h, int w, int co, int cin, int dtype, float eps, void* stream);

/* ---- weights ---------------------------------------------------------------------------------- */
/* bytes of the packed image of a weight with `rows` GEMM rows (any count; padded to 16 inside),
 * `c_pad` k-side channels (multiple of 8) and ntaps taps (27, 8, or 1 for VS_PACK_SCATTER_D1). */
size_t vs_packed_weight_bytes(int rows, int c_pad, int ntaps, int dtype);
/* src: float[d0][d1][ntaps]; c_pad >= the k-side channel count, zero-filled beyond it. */
int vs_pack_weight(const float* src, void* dst, int d0, int d1, int ntaps, int c_pad, int form, int dtype, void* stream);

/* One launch packing many weights (all trainable convs of a network after an optimiser step): descs is a DEVICE array. */
typedef struct vs_pack_desc {
    const float* src;      /* float[d0][d1][ntaps] */
    void* dst;             /* packed image, vs_packed_weight_bytes() bytes */
    int d0, d1, ntaps, c_pad, form, dtype;
    int first_block;       /* index of this weight's first 256-thread block in the launch; a thread packs one 16-byte fragment (8 bf16 / 4 fp32 elements) */
    int pad_;
    long long total;       /* packed elements of this weight */
} vs_pack_desc;
int vs_pack_weight_multi(const vs_pack_desc* descs, int n_desc, int total_blocks, void* stream);

/* ---- convolutions (implicit GEMM on MFMA) ------------------------------------------------------ */
/* y[n,v,m] = bias[m] + sum_{tap,c} act(x)[n, v*s + off(tap) - p, c] * W[m][tap][c]
 *   kind K3  : x,y both (N,D,H,W);   kind K2S2: x is (N,D,H,W), y is (N,D/2,H/2,W/2).
 * Also the backward-data of K3 (with VS_PACK_ROWS_D1_FLIP weights) and of T2S2 (kind K2S2 with the
 * transposed-conv weight packed VS_PACK_ROWS_D0).
 *   x_stats  : lazy-activation stats of x or NULL         y_stats : if non-NULL, (sum,sumsq) of y are
 *   bias     : float[m_out] or NULL                                 ACCUMULATED into it (caller zeroes)
 *   c_in     : channels of x (multiple of 8)   m_out : channels of y (multiple of 8; rows beyond the
 *              real weight rows come out as bias/zero)
 */
int vs_conv_gather_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                       void* y, double* y_stats, int n, int d, int h, int w, int c_in, int m_out,
                       int kind, int dtype, float eps, void* stream);

/* y[n, 2v+off(tap), m] = bias[m] + sum_c act(x)[n,v,c] * W[tap][m][c]      (x is (N,D,H,W), y (N,2D,2H,2W))
 * ConvTranspose3d(k2,s2) forward (joint_model.py:118) and Conv3d(k2,s2) backward-data. */
int vs_conv_scatter_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                        void* y, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps, void* stream);

/* Backward-data forms of the two calls above with the InstanceNorm+ReLU-backward REDUCTION fused into the epilogue:
 * y here is g = dL/da of a lazy activation a = relu(instnorm(mask_x)) (mask_x: raw tensor shaped like y, mask_stats
 * its (sum,sumsq)); besides writing g the kernel ACCUMULATES sums[n][m] = (sum g*[xhat>0], sum g*[xhat>0]*xhat)
 * (caller zeroes), i.e. exactly what vs_instnorm_relu_bwd_reduce(g, mask_x, mask_stats) would return, without re-reading g. */
int vs_conv_gather_bwd_data(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats,
                            double* sums, int n, int d, int h, int w, int c_in, int m_out, int kind, int dtype, float eps,
                            void* stream);
int vs_conv_scatter_bwd_data(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats,
                             double* sums, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps, void* stream);
/* vs_conv_gather_bwd_data (K3) whose INPUT gradient arrives un-applied: g = dL/da of the lazy activation a = relu(instnorm(act_x)), with
 * that activation's statistics (act_stats) and IN-backward sums (act_sums, complete: produced by the backward-data launch that wrote g).
 * The apply pass rstd * (g*[xhat>0] - m1 - xhat*m2) runs while the halo tile is staged — the standalone vs_instnorm_relu_bwd_apply launch
 * between two backward-data launches of the 8-channel full-resolution layers (3 tensor passes at 96^3) disappears; dx_out (nullable)
 * receives the applied gradient (what the weight gradient of this layer reads).  mask_x / mask_stats / sums: all three as in
 * vs_conv_gather_bwd_data (the conv's own input is a lazy activation) or all NULL (it is a stored tensor: no sums to accumulate).
 * 16-bit storage; kernels exist for the single-chunk layers of the full- and half-resolution levels (c_in 8 or 16: igemm_k3t.h, igemm_k3b.h FA);
 * any other shape returns VS_ESHAPE — ask vs_conv_k3_fused_apply_supported first (1 / 0; lazy_input: the conv's own input is a lazy activation). */
int vs_conv_k3_fused_apply_supported(int n, int d, int h, int w, int c_in, int m_out, int lazy_input, int dtype);
/* Backward-data of a 3x3x3 conv whose own input is a lazy activation (vs_conv_gather_bwd_data, K3) WITH THE LAYER'S WEIGHT GRADIENT in the same launch
 * (autograd of joint_model.py:40-48 at full resolution; csrc/igemm_k3tw.h): the launch already holds both operands of dW — the applied output gradient as
 * its halo tile, relu(instnorm(mask_x)) under every output voxel for the fused sums — so neither is read again by the grouped weight-gradient launch.
 * act_x / act_stats / act_sums: all three (g arrives un-applied, as for vs_conv_k3_bwd_data_fused_apply; the applied gradient is then never stored) or all NULL.
 * slabs: vs_conv_k3_bwd_data_wgrad_slabs(n, d, h, w) partial sums float[27][8][8] each, S[o][c][m] = sum_v Q(v)[c] g_applied(v + o)[m] = dW[m][c][-o];
 * hand them to vs_conv_wgrad_multi as a VS_WGRAD_SLABS descriptor (fixed summation order: bitwise reproducible).
 * 16-bit storage, c_in = m_out = 8 stored channels (the 96^3 / 128^3 / 160^3 levels); vs_conv_k3_bwd_data_wgrad_supported says 1 / 0 (VS_FUSE_WGRAD=0: always 0). */
int vs_conv_k3_bwd_data_wgrad_supported(int n, int d, int h, int w, int c_in, int m_out, int dtype);
int vs_conv_k3_bwd_data_wgrad_slabs(int n, int d, int h, int w);
int vs_conv_k3_bwd_data_wgrad(const void* g, const void* act_x, const double* act_stats, const double* act_sums, const void* w_packed, void* y,
                              const void* mask_x, const double* mask_stats, double* sums, float* slabs, int n, int d, int h, int w, int c_in,
                              int m_out, int dtype, float eps, void* stream);
/* out_block's backward (two classes, 8 stored channels; joint_model.py:224-225,265-266 / 366-367,386-388) as ONE launch: the gradient of the logits is formed
 * from the planar probabilities and their gradient(s) while the tile is staged — exactly vs_softmax2_cl_bwd(prob, gprob, gprob_cl), logit dropout included,
 * never stored — and multiplied out as vs_conv_gather_bwd_data would (y, fused sums against mask_x / mask_stats).  slabs (nullable): the layer's weight
 * gradient as for vs_conv_k3_bwd_data_wgrad; bias_part (nullable, needs slabs): one (sum gl0, sum gl1) pair of doubles per slab — give both to
 * vs_conv_wgrad_multi as ONE VS_WGRAD_SLABS descriptor (bias_g = bias_part, bias_rows = the slab count, bias_c_real = 2, db).  Shapes: those of
 * vs_conv_k3_bwd_data_wgrad_supported(n, d, h, w, 8, 8, dtype). */
int vs_conv_k3_softmax2_bwd_data(const float* prob, const float* gprob, const void* gprob_cl, const void* w_packed, void* y, const void* mask_x,
                                 const double* mask_stats, double* sums, float* slabs, double* bias_part, int n, int d, int h, int w, int dtype,
                                 float eps, float drop_p, unsigned long long drop_seed, void* stream);
/* The library's tuning switches, in ONE place (round 6; csrc/config.hip).  The library keeps one process-wide copy: filled on first use from the environment
 * variables named below (so A/B scripts that set them before the process starts keep working), read by every launcher on EVERY call (nothing is cached in
 * function-local statics), replaced as a whole by vs_set_config() — the only writer; set it between launches, not concurrently with one.  vs_config_bytes():
 * sizeof(vs_config) of the loaded library (bindings check their layout against it).  vs_config_from_env(): defaults + environment into *c without installing it. */
typedef struct vs_config {
    int k3_small;              /* VS_K3_SMALL 1: volumes up to 6^3 with C % 32 == 0 run k3s_kernel (whole padded sample in LDS) */
    int k3_tall;               /* VS_K3_TALL -1: 4x8x16 tiles where measured faster (auto); 0 / 1 force */
    int k3_wgs_per_cu;         /* VS_K3_WGS_PER_CU 0: persistent workgroups per CU of the 3x3x3 kernels (0 = per-kernel default: 3 for k3b, 4 for the exact-f32 k3) */
    int k3t_wgs_per_cu;        /* VS_K3T_WGS_PER_CU 2: the 8-channel full-resolution kernels (k3t, k3tw) */
    int k3f_min_wgs;           /* VS_K3F_MIN_WGS 512: exact-f32 C >= 32 layers take shorter tiles until the launch has this many workgroups */
    int mt_min_wgs;            /* VS_MT_MIN_WGS 1024: largest row tile that still leaves this many workgroups */
    int f32_limbs;             /* VS_F32_LIMBS 1: fp32 parity mode on the bf16 matrix cores (three-limb operands); 0 = exact-f32 MFMA everywhere */
    int g1_limbs;              /* VS_G1_LIMBS 1: ... in the stride-2 / transposed kernels */
    int k3x_ck;                /* VS_K3X_CK 8: channel chunk of the limb kernels (8 or 16) */
    int k3x_toeplitz;          /* VS_K3X_TOEPLITZ 1 */
    int fuse_wgrad;            /* VS_FUSE_WGRAD 1: vs_conv_k3_bwd_data_wgrad_supported may say 1 */
    int epilogue_apply;        /* VS_EPILOGUE_APPLY 1: vs_conv_k3_bwd_data_applied_supported may say 1 */
    int chain;                 /* VS_CHAIN 1: vs_conv_k3_chain_supported may say 1 */
    int k2s2_stream;           /* VS_K2S2_STREAM 1: the streaming kernel for the full-resolution stride-2 backward-data scatter */
    int k2s8_wgs_per_cu;       /* VS_K2S8_WGS_PER_CU 4 */
    int up_wgs_per_cu;         /* VS_UP_WGS_PER_CU 2: composed Up head kernels */
    int up_rb;                 /* VS_UP_RB 0: row blocks per workgroup of the composed Up forward (0 = chosen per shape) */
    int wgrad_uber;            /* VS_WGRAD_UBER 1: every bucket of a grouped weight-gradient pass in one grid */
    int wgrad_mpack;           /* VS_WGRAD_MPACK 1: M-packed rows for layers with 8 stored output channels */
    int wgrad_swap;            /* VS_WGRAD_SWAP 1: operand exchange for lazy-input 3x3x3 layers */
    int wgrad_big;             /* VS_WGRAD_BIG 1: 8x8x16 tiles for the full-resolution layers */
    int wgrad_xcd;             /* VS_WGRAD_XCD 2: a layer's workgroups re-ranked so that one XCD walks neighbouring tiles and holds all channel-block pairs of a tile (1: single-pair layers only; 0: plain order) */
    int k3_short_tiles;        /* VS_K3_SHORT_TILES 2: 4x2x16 tiles for the 32-channel 3x3x3 launches of at most 128 workgroups (the 12^3-class levels), 4x1x16 where that still leaves at most 128 (2); 1: 4x2x16 only; 0: off */
    int wgrad_bias_fold;       /* VS_WGRAD_BIAS_FOLD 1: a ConvTranspose3d's bias gradient is summed by its weight-gradient workgroups (the same tensor is their Q operand) */
    long long wgrad_wgs;            /* VS_WGRAD_WGS 512: workgroups per ungrouped weight-gradient launch */
    long long wgrad_f32_tiles;      /* VS_WGRAD_F32_TILES 8 */
    long long wgrad_group_wgs;      /* VS_WGRAD_GROUP_WGS 0: workgroups per bucket of a grouped pass (0 = chosen per pass) */
    long long wgrad_big_min_voxels; /* VS_WGRAD_BIG_MIN_VOXELS 400000 */
} vs_config;
int vs_config_bytes(void);
int vs_config_from_env(vs_config* c);
int vs_get_config(vs_config* out);
int vs_set_config(const vs_config* cfg);
/* vs_conv_gather_bwd_data (K3) FOLLOWED BY vs_instnorm_relu_bwd_apply of its output, in one launch (csrc/igemm_k3b.h EA, round 6): every workgroup keeps its tile's
 * rounded outputs in registers, adds its partial IN-backward sums, arrives on its SAMPLE's counter (the statistics of joint_model.py:11's InstanceNorm3d are per
 * sample: nothing of another sample is waited for), reads the complete sums back and stores y = dL/d(raw tensor) — the un-applied gradient is never written and the
 * standalone apply launch disappears.  Results equal the two launches' bit for bit in the deterministic build.  sync: n * 1024 ZEROED bytes (8 counter shards per sample), 128-byte aligned,
 * private to the call; fault: the device word of vs_conv_k3_chain.  16-bit storage, c_in a multiple of 32, volumes above 6^3 whose launch is one resident round
 * (at most 512 workgroups: the 24^3 / 12^3 levels): vs_conv_k3_bwd_data_applied_supported says 1 / 0 (env VS_EPILOGUE_APPLY=0: always 0). */
int vs_conv_k3_bwd_data_applied_supported(int n, int d, int h, int w, int c_in, int m_out, int dtype);
int vs_conv_k3_bwd_data_applied(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums,
                                unsigned int* sync, unsigned int* fault, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps, void* stream);
/* The same for the stride-2 kinds (csrc/igemm.h g1_kernel, round 6): scatter = 0: vs_conv_gather_bwd_data(kind K2S2) — the backward-data of
 * nn.ConvTranspose3d(C, C, 2, stride=2) (joint_model.py:118) — scatter = 1: vs_conv_scatter_bwd_data — the backward-data of nn.Conv3d(C, C, 2, stride=2)
 * (joint_model.py:130) — FOLLOWED BY vs_instnorm_relu_bwd_apply_add of its output, in one launch.  add (nullable, y's shape and type): a second gradient of the same raw
 * tensor (the U-Net skip's, joint_model.py:380,382), summed in after the apply as vs_instnorm_relu_bwd_apply_add does.  Any storage type; launches of at most 256
 * workgroups (the <= 48^3 levels at batch 2).  sync / fault / concurrency rule: as vs_conv_k3_bwd_data_applied.  Results equal the two launches' bit for bit in the
 * deterministic build. */
int vs_conv_s2_bwd_data_applied_supported(int n, int d, int h, int w, int c_in, int m_out, int scatter, int dtype);
int vs_conv_s2_bwd_data_applied(const void* x, const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums, const void* add,
                                unsigned int* sync, unsigned int* fault, int n, int d, int h, int w, int c_in, int m_out, int scatter, int dtype, float eps,
                                void* stream);
/* One DoubleConv (joint_model.py:35-52: three times [Conv3d 3x3x3 pad 1 -> InstanceNorm3d -> ReLU]) at the small volumes of the deep levels as ONE launch
 * (csrc/chain.h, round 6).  InstanceNorm3d is per (sample, channel) (joint_model.py:11), so layer l + 1 of sample n depends on layer l of sample n only: the
 * workgroups of a sample hand the raw output and its statistics over inside the launch (write-through stores, counter, sc1 loads) instead of ending it.
 * forward (backward = 0): layers[0..n_layers) in order; layer l = vs_conv_gather_fwd(K3) of (x, x_stats) -> (y, y_stats); for l > 0, x / x_stats ARE layer l-1's
 *   y / y_stats; layer 0's x_stats may be NULL (a stored input).
 * backward (backward = 1): layers in BACKWARD order; layer l = vs_conv_gather_bwd_data(K3) of the gradient x (layer 0: w.r.t. the block's raw output, applied;
 *   l > 0: layer l-1's y) with the transposed / mirrored weight image -> y = dL/d(activation), sums against (mask_x, mask_stats) = that activation's raw tensor and
 *   statistics; apply = 1: vs_instnorm_relu_bwd_apply runs on y IN PLACE inside the launch (y leaves as dL/d(raw tensor): what the next layer and the weight gradient read).
 *   Every layer but the last must apply; the last may have mask_x = mask_stats = sums = NULL (the block's input is a stored tensor: plain backward-data).
 *   add (nullable; needs the last layer's apply): a second gradient of the last layer's raw tensor, summed in as vs_instnorm_relu_bwd_apply_add does.
 * sync: vs_conv_k3_chain_sync_bytes(n) ZEROED bytes, 128-byte aligned, private to this call; fault: a device word the kernel ORs 1 into when a bounded wait gave up
 *   (never in a correct launch: the library rejects chains whose workgroups cannot all be resident) — results are then invalid, the queue is not hung.
 * CONCURRENCY: the workgroups of a sample wait for one another inside the launch, so all of them must be resident together.  The library guarantees that for a
 *   launch that is not competing for CUs with ANOTHER launch of this kind (vs_conv_k3_chain, vs_conv_k3_bwd_data_applied): one process per GPU issuing its step on
 *   one stream — the deployment model.  Two such launches running at the same time on one device (two processes sharing a GPU, two streams) can each hold part of
 *   the chip and starve the other's waiters; the bounded waits then give up and raise `fault`.  A host that shares a device switches the two forms off first
 *   (vs_set_config: chain = 0, epilogue_apply = 0; Python: ops.device_is_shared()).
 * Shapes: (d+2)(h+2)(w+2) <= 512 (up to 6^3), channels multiples of 32 (c_in) / 8 (m_out), at most 256 workgroups per sample (32 per XCD): vs_conv_k3_chain_supported(n, d, h, w,
 * largest channel count of the chain, dtype) says 1 / 0 (env VS_CHAIN=0: always 0).  All three storage types; results are bit-identical to the per-layer launches
 * in the deterministic build. */
typedef struct vs_chain_layer {
    const void* x;
    const double* x_stats;
    const void* w_packed;
    void* y;
    double* y_stats;
    const void* mask_x;
    const double* mask_stats;
    double* sums;
    int c_in, m_out;
    int apply;
    int reserved_;
} vs_chain_layer;
int vs_conv_k3_chain_supported(int n, int d, int h, int w, int c_max, int dtype);
long long vs_conv_k3_chain_sync_bytes(int n);
int vs_conv_k3_chain(const vs_chain_layer* layers, int n_layers, int backward, const void* add, unsigned int* sync, unsigned int* fault,
                     int n, int d, int h, int w, int dtype, float eps, void* stream);
/* fp32 parity mode: 1 when a 3x3x3 convolution of dtype VS_F32 on a (d, h, w) volume with c_in stored input channels runs on the bf16 matrix cores through exact three-limb operand splitting
 * (csrc/igemm_k3x.h: every fp32 operand = three bf16 limbs, six exact limb products per product, fp32 accumulation — 2.7x fewer matrix cycles
 * than the exact-f32 MFMA at the accuracy of one fp32 rounding).  Their packed weights must then be VS_F32X3 images: vs_pack_weight /
 * vs_packed_weight_bytes / vs_pack_weight_multi with dtype VS_F32X3 (a PACK-ONLY dtype: tensors stay VS_F32).  Env VS_F32_LIMBS=0 -> 0: the
 * exact-f32 MFMA kernels with plain VS_F32 images; so do the volumes up to 6^3 with c_in a multiple of 32 (k3s_kernel<float>). */
int vs_conv_k3_f32_limbs(int d, int h, int w, int c_in);

int vs_conv_k3_bwd_data_fused_apply(const void* g, const void* act_x, const double* act_stats, const double* act_sums,
                                    const void* w_packed, void* y, const void* mask_x, const double* mask_stats, double* sums,
                                    void* dx_out, int n, int d, int h, int w, int c_in, int m_out, int dtype, float eps, void* stream);

/* out_block + Softmax(dim=1) fused (joint_model.py:224-225,265-266,366-367,386-388):
 * prob[n][k][v] (planar fp32, k < 2) = softmax_k( bias[k] + conv3x3x3(act(x))[k] ). */
int vs_conv_k3_softmax2_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                            float* prob, int n, int d, int h, int w, int c_in, int dtype, float eps, void* stream);
/* same with F.dropout applied to the two logits before the softmax (Segmentation.forward's last dropout site,
 * joint_model.py:386-388); drop_p == 0 is the call above. */
int vs_conv_k3_softmax2_dropout_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                                    float* prob, int n, int d, int h, int w, int c_in, int dtype, float eps, float drop_p,
                                    unsigned long long drop_seed, void* stream);
/* same, additionally writing the probabilities as a channels-last bf16 tensor prob_cl[n][v][8] (2 real channels) — the input layout of the
 * network that consumes them next (Joint.forward feeds Segmentation's prediction to the VAE, joint_model.py:447-450): saves the
 * vs_pack_planar launch and its re-read.  prob_cl may be NULL (then this is the call above); bf16 only. */
int vs_conv_k3_softmax2_cl_fwd(const void* x, const double* x_stats, const void* w_packed, const float* bias,
                               float* prob, void* prob_cl, int n, int d, int h, int w, int c_in, int dtype, float eps, float drop_p,
                               unsigned long long drop_seed, void* stream);

/* weight gradient of all three conv kinds:
 *   dW[m][c][tap] = sum_{n,v} actP(P)[n,v,m] * actQ(Q)[n, v*s + off(tap) - p, c]        (fp32, reference layout)
 * Conv3d (K3/K2S2): P = dL/dy on the OUTPUT grid (dp,hp,wp), Q = the conv input  -> dW[co][ci][tap].
 * ConvTranspose3d : P = the (coarse) input on (dp,hp,wp), Q = dL/dy (fine grid), kind = VS_CONV_K2S2 -> dW[ci][co][tap].
 * m_real / c_real: how many leading channels of P / Q are real (dW has exactly m_real*c_real*ntaps floats).
 * workspace: vs_conv_wgrad_workspace_bytes() bytes, contents undefined on entry. */
size_t vs_conv_wgrad_workspace_bytes(int n, int dp, int hp, int wp, int m_ch, int c_ch, int kind);
int vs_conv_wgrad(const void* p, const double* p_stats, const void* q, const double* q_stats, float* dw,
                  void* workspace, size_t workspace_bytes, int n, int dp, int hp, int wp, int m_ch, int c_ch,
                  int m_real, int c_real, int kind, int dtype, float eps, void* stream);
/* The weight (and bias) gradients of MANY conv layers in a handful of launches.  They are leaves of backward — nothing in
 * the pass reads them (the optimiser step of main_source.py:660-661 does, after backward) — so the host side collects one
 * descriptor per layer while autograd walks the graph and issues them together when the pass ends: layers of the same
 * kernel instantiation share one grid, all slab reductions share one.  Per layer the arguments mean what they mean in
 * vs_conv_wgrad; bias_g != NULL additionally requests db[c] = sum over bias_rows of bias_g[bias_rows][bias_c_ch],
 * c < bias_c_real (fixed summation order: bitwise reproducible, unlike vs_bias_grad's float atomics).
 * Descriptors that share `dw` (and `db`) are the uses of ONE weight in this backward pass (a network applied several times, e.g. the VAE
 * inside Embed, joint_model.py:469-500): their contributions are summed — the slabs of all uses feed one reduction — as autograd's
 * accumulation would, without an add launch per use.
 * workspace: vs_conv_wgrad_multi_workspace_bytes() bytes for the same (descs, count, dtype), contents undefined. */
typedef struct vs_wgrad_desc {
    const void* p; const double* p_stats;
    const void* q; const double* q_stats;
    float* dw;
    const void* bias_g; float* db;
    long long bias_rows;
    int bias_c_ch, bias_c_real;
    int n, dp, hp, wp, m_ch, c_ch, m_real, c_real, kind;
    int reserved_;      /* VS_CONV_UP: Co, the channels of the fine tensor p points to; 0 otherwise */
} vs_wgrad_desc;
size_t vs_conv_wgrad_multi_workspace_bytes(const vs_wgrad_desc* descs, int count, int dtype);
int vs_conv_wgrad_multi(const vs_wgrad_desc* descs, int count, void* workspace, size_t workspace_bytes, int dtype,
                        float eps, void* stream);
/* db[c] = sum over rows of g[rows][c_ch], c < c_real (bias gradient of a conv whose bias is live). */
int vs_bias_grad(const void* g, float* db, long long rows, int c_ch, int c_real, int dtype, void* stream);
/* same; accumulate != 0 adds to db instead of overwriting it (a bias used several times in one backward pass) */
int vs_bias_grad_acc(const void* g, float* db, long long rows, int c_ch, int c_real, int dtype, int accumulate, void* stream);

/* ---- InstanceNorm3d + ReLU ---------------------------------------------------------------------- */
/* (sum,sumsq) per (n,c) of x, accumulated into stats (caller zeroes) — only needed when the producer
 * of x was not one of the convs above. */
int vs_instnorm_stats(const void* x, double* stats, int n, long long voxels, int c, int dtype, void* stream);
/* a = relu((x-mean)*rstd) [+ relu((x2-mean2)*rstd2)]  — materialises a lazy activation; with the second
 * operand it is the U-Net skip `up(x)+x3` (joint_model.py:380,382). */
int vs_instnorm_relu_fwd(const void* x, const double* x_stats, const void* x2, const double* x2_stats,
                         void* out, int n, long long voxels, int c, int dtype, float eps, void* stream);
/* backward of a = relu(instnorm(x)) given g = dL/da:
 *   reduce: sums[n][c] = (sum g*[xhat>0], sum g*[xhat>0]*xhat)   (ACCUMULATED, caller zeroes)
 *   apply : gx = rstd * (g*[xhat>0] - mean_v(.) - xhat * mean_v(. * xhat))                                      */
int vs_instnorm_relu_bwd_reduce(const void* g, const void* x, const double* x_stats, double* sums,
                                int n, long long voxels, int c, int dtype, float eps, void* stream);
int vs_instnorm_relu_bwd_apply(const void* g, const void* x, const double* x_stats, const double* sums,
                               void* gx, int n, long long voxels, int c, int dtype, float eps, void* stream);
/* apply with a second gradient of the same raw tensor summed in: gx = round(apply(g)) + add — an activation that feeds both the next
 * encoder level and a U-Net skip (joint_model.py:380,382) receives two gradients; autograd would add them with a kernel of its own. */
int vs_instnorm_relu_bwd_apply_add(const void* g, const void* x, const double* x_stats, const double* sums, const void* add,
                                   void* gx, int n, long long voxels, int c, int dtype, float eps, void* stream);
/* p[0 .. bytes) = 0 (16-byte aligned, bytes a multiple of 16): the per-forward statistics arena (every (sum, sumsq) / IN-backward-sums
 * buffer of a pass is carved from ONE zeroed allocation). */
int vs_zero_fill(void* p, long long bytes, void* stream);
/* The backward of the additive skip a = relu(instnorm(x1)) + relu(instnorm(x2)) (joint_model.py:380,382): one gradient g, both
 * operands' reduce in one launch and both applies in one (sums1 / sums2 ACCUMULATED: caller zeroes; gx1 / gx2 distinct from g). */
int vs_instnorm_relu_bwd_pair(const void* g, const void* x1, const double* x1_stats, double* sums1, void* gx1,
                              const void* x2, const double* x2_stats, double* sums2, void* gx2, int n, long long voxels,
                              int c, int dtype, float eps, void* stream);

/* ---- general normalisation + activation ------------------------------------------------------------------------------------------
 * The settings of joint_model.py's blocks that no entry point of the reference passes: norm_type=2 = nn.BatchNorm3d(C, momentum=0.1)
 * (joint_model.py:12-13, the constructors' default) with its affine pair and running statistics, and soft=True = torch.nn.Softplus()
 * (joint_model.py:38,58,93,104).  Not fused into the convs: the conv writes its raw output and (sum, sumsq) statistics as always and
 * these streaming passes follow; the activation that leaves is a stored tensor.  u = xhat * gamma + beta, xhat = (x - mean) * rstd. */
enum { VS_NORM_INSTANCE = 0, VS_NORM_BATCH = 1, VS_NORM_BATCH_EVAL = 2, VS_NORM_NONE = 3 };   /* per (n,c) | pooled over the batch (training) | running statistics | no normalisation: mean 0, rstd 1 (conv -> activation of the *_GS blocks, joint_model.py:58-63) */
enum { VS_ACT_RELU = 0, VS_ACT_SOFTPLUS = 1 };                               /* Softplus: beta 1, threshold 20 (torch defaults) */
/* mean / rstd tables float[n][c] from a conv epilogue's statistics (double[VS_STAT_SLOTS][n][c][2], `count` voxels per sample).
 * VS_NORM_BATCH: batch mean and biased variance; running_mean / running_var (nullable pair, fp32 [c_real]) take
 * (1-momentum) * old + momentum * (mean | unbiased variance) and *num_batches_tracked (nullable) is incremented — torch.nn.BatchNorm3d
 * in training mode.  VS_NORM_BATCH_EVAL: the tables are the running pair (stats unused). */
int vs_norm_tables(const double* stats, int n, int c, int c_real, double count, int mode, float eps, float momentum,
                   float* running_mean, float* running_var, long long* num_batches_tracked, float* mean, float* rstd, void* stream);
/* y = act(u); channels >= c_real (padding) are written as zeros.  gamma / beta: fp32 [c_real] or NULL (1 / 0). */
int vs_norm_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, int n,
                    long long voxels, int c, int c_real, int act, int dtype, void* stream);
/* backward, given g = dL/dy:  da = g * act'(u)
 *   reduce: sums[n][c] = (sum da, sum da * xhat)          (double[VS_STAT_SLOTS][n][c][2], ACCUMULATED: caller zeroes)
 *   finish: coef[n][c] = (k, m1, m2) with k = gamma * rstd and (m1, m2) = the means of (da, da * xhat) over the sample (INSTANCE),
 *           over the batch (BATCH) or 0 (BATCH_EVAL);  dgamma[c] = sum_n sum da * xhat, dbeta[c] = sum_n sum da (nullable)
 *   apply : dx = k * (da - m1 - xhat * m2) */
int vs_norm_act_bwd_reduce(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           double* sums, int n, long long voxels, int c, int c_real, int act, int dtype, void* stream);
int vs_norm_act_bwd_finish(const double* sums, int n, int c, int c_real, double count, int mode, const float* rstd, const float* gamma,
                           float* coef, float* dgamma, float* dbeta, void* stream);
int vs_norm_act_bwd_apply(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* coef, void* dx, int n, long long voxels, int c, int c_real, int act, int dtype, void* stream);

/* ---- the `*_GS` model family (joint_model.py:17-33, 54-99, 307-346; instantiated nowhere in the reference) ------------------------------
 * GSNorm3d (joint_model.py:17-33): y[c] = x[c] / (sum of x over c's group of c / num_group consecutive channels + 1e-4), per voxel;
 * rows = n * voxels channels-last rows of c channels.  bwd: dx[c] = g[c] / S - (sum_j g[j] x[j]) / S^2. */
int vs_gsnorm_fwd(const void* x, void* y, long long rows, int c, int num_group, int dtype, void* stream);
int vs_gsnorm_bwd(const void* g, const void* x, void* dx, long long rows, int c, int num_group, int dtype, void* stream);
/* torch.nn.Upsample(scale_factor=scale, mode='trilinear') (joint_model.py:69,323-325; align_corners=False): x (n,d,h,w,c) -> y (n,d*s,h*s,w*s,c).
 * bwd: g (upsampled grid) -> dx; scratch = n*d*h*w*c floats (16-byte aligned; zeroed and used for fp32 atomic accumulation by the call). */
int vs_upsample_trilinear_fwd(const void* x, void* y, int n, int d, int h, int w, int c, int scale, int dtype, void* stream);
int vs_upsample_trilinear_bwd(const void* g, void* dx, float* scratch, int n, int d, int h, int w, int c, int scale, int dtype, void* stream);
/* nn.Softmax(dim=1) over two classes as its own pass (Segmentation_GS.final after the 1x1x1 out_block2, joint_model.py:326-327,343-344):
 * channels 0, 1 of channels-last logits -> planar fp32 [n][2][voxels]; its backward is vs_softmax2_bwd. */
int vs_softmax2_fwd(const void* logits, float* prob, int n, long long voxels, int c, int dtype, void* stream);

/* F.dropout(x, p, training=True) on a channels-last tensor (joint_model.py:256-264,379-385): out = x * keep / (1-p),
 * keep ~ Bernoulli(1-p) from a counter-based hash of (seed, element index) — the same call with the same seed applied to
 * the incoming gradient is the backward.  (The reference draws from torch's Philox stream; only the distribution matches.) */
int vs_dropout(const void* x, void* out, long long count, float p, unsigned long long seed, int dtype, void* stream);
/* The multiplier vs_dropout (and the fused logit dropout of vs_conv_k3_softmax2_dropout_fwd) applies to element i, i < count:
 * mask[i] = 0 or 1/(1-p).  Element order: the tensor's own memory order — channels-last [N][V][C] for vs_dropout, planar [N][2][V]
 * for the logits.  Lets a checker feed the SAME mask to the reference arithmetic (F.dropout replaced by a multiply). */
int vs_dropout_mask(float* mask, long long count, float p, unsigned long long seed, void* stream);

/* ---- training data pipeline on the device ---------------------------------------------------------------------------
 * Planar fp32 volumes [D][H][W].  What utils/utils.py's transform chain does per sample on CPU workers (main_source.py:191-211):
 * NumpyLoader_Multi_merge relabelling (utils.py:253-262), CropResize (utils.py:326-383), MySpatialTransform = batchgenerators
 * augment_spatial (utils.py:927-968; rotation, scale, random crop; cubic-spline image / nearest label), Clip + CenterIntensities
 * (utils.py:508-533, 575-618).  The interpolation arithmetic is scipy.ndimage's, which skimage.transform.resize and augment_spatial call. */
/* box6 = {min z, y, x, max z, y, x} of label > 0 (INT_MAX / -1 when there is none) */
int vs_data_bbox(const float* label, int d, int h, int w, int* box6, void* stream);
/* out = target of the last (source -> target) pair whose source equals the label, 0 otherwise; sources / targets: HOST arrays, n_pairs <= 16 */
int vs_data_relabel(const float* in, float* out, long long total, const float* sources, const float* targets, int n_pairs, void* stream);
/* dst[p] = src[p - off + lo] for p - off + lo inside [lo, hi) of the source, 0 elsewhere (crop + np.pad); lo3 / hi3 / off3: HOST arrays */
int vs_data_crop_pad(const float* src, float* dst, int sd, int sh, int sw, int dd, int dh, int dw, const int* lo3, const int* hi3,
                     const int* off3, void* stream);
/* scipy.ndimage.gaussian_filter1d(mode='mirror', truncate=4.0) along one axis (src != dst) */
int vs_data_gaussian_axis(const float* src, float* dst, int d, int h, int w, int axis, float sigma, void* stream);
int vs_data_minmax(const float* x, long long total, float* minmax2, void* stream);
/* scipy.ndimage.zoom(mode='mirror', grid_mode=True), order 0 or 1; order 1 results are clamped to [clip_lo, clip_hi] unless clip_lo > clip_hi */
int vs_data_zoom(const float* src, float* dst, int sd, int sh, int sw, int dd, int dh, int dw, int order, float clip_lo, float clip_hi, void* stream);
/* order-3 B-spline coefficients of x in fp64 (scipy.ndimage.spline_filter, mirror boundary) */
int vs_data_spline3_prefilter(const float* x, double* coef, int d, int h, int w, void* stream);
/* dst[o] = sample(src, A (o - (P-1)/2) + ctr), scipy map_coordinates(mode='constant', cval): order 3 (src = the fp64 coefficients) or 0 (src = fp32
 * volume); a9 (row-major 3x3) / ctr3: HOST arrays */
int vs_data_affine_sample(const void* src, float* dst, int sd, int sh, int sw, int pd, int ph, int pw, const double* a9, const double* ctr3, int order,
                          float cval, void* stream);
/* x = (clamp(x, lo, hi) - subtrahend) / divisor, in place */
int vs_data_clip_center(float* x, long long total, float lo, float hi, float subtrahend, float divisor, void* stream);

/* ---- layout glue at the NCDHW boundary ----------------------------------------------------------- */
/* planar fp32 [N][c_src][V] -> channels-last [N][V][c_pad] (zero-filled channels >= c_src) */
int vs_pack_planar(const float* src, void* dst, int n, long long voxels, int c_src, int c_pad, int dtype, void* stream);
/* channels-last [N][V][c_pad] -> planar fp32 [N][c_dst][V] */
int vs_unpack_planar(const void* src, float* dst, int n, long long voxels, int c_dst, int c_pad, int dtype, void* stream);
/* backward of the 2-class softmax: glogit[n,v,k] = p_k (g_k - sum_j p_j g_j), written channels-last with c_pad
 * channels (k >= 2 zero).  prob, gprob planar fp32 [N][2][V]. */
int vs_softmax2_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad, int dtype, void* stream);
/* same, followed by the backward of the logit dropout of vs_conv_k3_softmax2_dropout_fwd (same p / seed) */
int vs_softmax2_dropout_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad, int dtype,
                            float drop_p, unsigned long long drop_seed, void* stream);
/* same with the upstream gradient arriving in two parts: gprob (planar fp32, may be NULL) and gprob_cl (channels-last [n][v][c_pad] in
 * `dtype`, channels 0..1 used, may be NULL) — the gradients of prob and of its channels-last copy (vs_conv_k3_softmax2_cl_fwd). */
int vs_softmax2_cl_bwd(const float* prob, const float* gprob, const void* gprob_cl, void* glogit, int n, long long voxels,
                       int c_pad, int dtype, float drop_p, unsigned long long drop_seed, void* stream);
/* nn.Softmax(dim=1) over n_class = 1..8 classes as its own pass (joint_model.py:225,266 / 367,388 with a label set of more than one structure:
 * main_source.py:92-93 makes n_class = 1 + the number of --pan_index entries; the two-class case has the fused kernels above).
 * fwd: channels 0..n_class-1 of channels-last logits [n][v][c_pad] -> planar fp32 prob [n][n_class][v]; drop_p > 0 first applies
 * F.dropout to the logits (joint_model.py:386-387; element index = planar index into [n][n_class][v]).
 * bwd: glogit[n,v,k] = p_k (g_k - sum_j p_j g_j) (then the dropout's backward), written channels-last with c_pad channels, k >= n_class zero. */
int vs_softmax_cl_fwd(const void* logits, float* prob, int n, long long voxels, int c_pad, int n_class, int dtype, float drop_p,
                      unsigned long long drop_seed, void* stream);
int vs_softmax_cl_bwd(const float* prob, const float* gprob, void* glogit, int n, long long voxels, int c_pad, int n_class, int dtype,
                      float drop_p, unsigned long long drop_seed, void* stream);
/* label (float, values 0..n_class-1) [N][1][V] -> one-hot planar fp32 [N][n_class][V]   (main_source.py:449-451) */
int vs_onehot(const float* label, float* out, int n, long long voxels, int n_class, void* stream);
/* hard masks of the validation Dice (utils/evaluation.py:58-64: torch.argmax over channels -> scatter_ into zeros): x, out planar
 * (n, n_class, voxels) fp32; out[b][k][v] = (k == argmax_k' x[b][k'][v]), ties to the first maximal channel as torch.argmax; any n_class >= 1 */
int vs_hard_onehot(const float* x, float* out, int n, int n_class, long long voxels, void* stream);
/* mode 0: (a >= 0.5) ; mode 1: a>hi -> 1, a<lo -> 0, else a        (utils/evaluation.py:9-18) */
int vs_binarize(const float* a, float* out, long long count, int mode, float lo, float hi, void* stream);

/* ---- fully connected (VAE bottleneck, joint_model.py:216-218,242-243,248-253) -------------------- */
/* y[b][j] = act( bias[j] + sum_k W[j][k] * x[b][phys(k)] )  with phys(k) = (k % pv)*pc + k / pv when pc > 0:
 * x is a channels-last activation [B][pv voxels][pc channels] read in the reference's flatten order
 * (k = c*pv + v); pc == 0 means x is a plain float[B][K].  x_dtype applies to x; y, W, bias are fp32. */
int vs_linear_fwd(const void* x, int x_dtype, const float* wgt, const float* bias, float* y, int batch, int k_in,
                  int j_out, int pc, int pv, int relu, void* stream);
/* two layers on the same input (fc_mean and fc_std, joint_model.py:216-217,241-243) in one launch; arguments as vs_linear_fwd */
int vs_linear_fwd_pair(const void* x, int x_dtype, const float* w1, const float* b1, float* y1, int relu1, const float* w2,
                       const float* b2, float* y2, int relu2, int batch, int k_in, int j_out, int pc, int pv, void* stream);
/* y[b][phys(j)] = bias[j] + sum_k W[j][k] * z[b][k]   — fc2: fp32 latent in, channels-last activation out. */
int vs_linear_fwd_perm_out(const float* z, const float* wgt, const float* bias, void* y, int y_dtype, int batch,
                           int k_in, int j_out, int pc, int pv, void* stream);
/* backward of vs_linear_fwd: gy is float[B][J] (already masked by the caller's ReLU if any).
 * gx (may be NULL) gets dL/dx in x's layout/dtype; gw / gb (may be NULL) are fp32 [J][K] / [J]. */
int vs_linear_bwd(const void* x, int x_dtype, const float* wgt, const float* gy, const float* y_for_relu,
                  void* gx, float* gw, float* gb, int batch, int k_in, int j_out, int pc, int pv, void* stream);
/* backward of vs_linear_fwd_perm_out: gy channels-last; gz float[B][K] (may be NULL); gw,gb may be NULL. */
int vs_linear_perm_out_bwd(const float* z, const float* wgt, const void* gy, int y_dtype, float* gz, float* gw,
                           float* gb, int batch, int k_in, int j_out, int pc, int pv, void* stream);

/* ---- VAE latent ---------------------------------------------------------------------------------- */
/* z = mean + noise*std*scale  (joint_model.py:248) ; bwd: gmean = gz, gstd = gz*noise*scale */
int vs_reparam_fwd(const float* mean, const float* std_, const float* noise, float scale, float* z, long long count, void* stream);
int vs_reparam_bwd(const float* gz, const float* noise, float scale, float* gmean, float* gstd, long long count, void* stream);
/* out = mean_b 0.5*(sum std^2 + sum mean^2 - 2 sum log(std+1e-5))     (utils/evaluation.py:42-45) */
int vs_kl_fwd(const float* mean, const float* std_, float* out, int batch, int dim, void* stream);
int vs_kl_bwd(const float* mean, const float* std_, const float* gout, float* gmean, float* gstd, int batch, int dim, void* stream);

/* ---- losses ---------------------------------------------------------------------------------------- */
/* soft Dice (utils/evaluation.py:48-80; main_source.py:150-182):  s,t planar fp32 [B][C][V].
 *   sums[b][c] = (sum s*t, sum s, sum t) for bot <= c < top (double[B][C][3], fully overwritten)
 *   per_sample[b] = mean_{c in [bot,top)} 2*I/(S+T+eps) ;  mean_out[0] = mean_b per_sample[b]            */
int vs_dice_fwd(const float* s, const float* t, double* sums, float* per_sample, float* mean_out,
                int batch, int channels, long long voxels, int bot, int top, float eps, void* stream);
/* gs/gt (either may be NULL) = d(sum_b w[b]*per_sample[b])/d(s|t); channels outside [bot,top) get 0.
 * gout_is_mean == 0: w[b] = gout[b] (float[B], upstream gradient of each per-sample score);
 * gout_is_mean == 1: w[b] = gout[0]/B (upstream gradient of mean_out). */
int vs_dice_bwd(const float* s, const float* t, const double* sums, const float* gout, int gout_is_mean, float* gs,
                float* gt, int batch, int channels, long long voxels, int bot, int top, float eps, void* stream);
/* A weighted sum of soft-Dice LOSSES of one source against k <= 4 targets in one pass over the source — the loss line of
 * every train method, e.g. main_source.py:469-471  final = lambda*(1-Dice(pred,recon)) + (1-Dice(pred,gt)):
 *   terms[j] = 1 - mean_b mean_{c in [bot,top)} 2*I_j/(S+T_j+eps);   final = ((w[0]*terms[0]) + w[1]*terms[1]) + ...   (fp32, that order)
 * t, w (and gt below) are HOST arrays of k entries.  scratch: vs_dice_loss_multi_scratch_doubles() doubles, contents undefined
 * on entry (per-block partials summed in a fixed order: no atomics, bitwise reproducible); the forward leaves the sums the
 * backward needs in its first 3*k*B*C entries.  The backward writes gs = d final/d s * gout[0] and,
 * for every non-NULL gt[j], d final/d t_j * gout[0]; channels outside [bot,top) get 0. */
size_t vs_dice_loss_multi_scratch_doubles(int k, int batch, int channels);
int vs_dice_loss_multi_fwd(const float* s, const float* const* t, const float* w, int k, double* scratch, float* terms,
                           float* final_out, int batch, int channels, long long voxels, int bot, int top, float eps, void* stream);
int vs_dice_loss_multi_bwd(const float* s, const float* const* t, const float* w, int k, const double* scratch,
                           const float* gout, float* gs, float* const* gt, int batch, int channels, long long voxels,
                           int bot, int top, float eps, void* stream);
/* The same with LABEL targets: labels[j] non-NULL (labels itself may be NULL) makes target j the one-hot of that label volume
 * ([batch][voxels] floats, truncated to int as vs_onehot does: main_source.py:449-451) evaluated on the fly — t[j] is then ignored,
 * gt[j] must be NULL, and the one-hot tensor (2 x the prediction's size, written once and read twice per step) never exists. */
int vs_dice_loss_multi_labels_fwd(const float* s, const float* const* t, const float* const* labels, const float* w, int k, double* scratch,
                                  float* terms, float* final_out, int batch, int channels, long long voxels, int bot, int top, float eps,
                                  void* stream);
int vs_dice_loss_multi_labels_bwd(const float* s, const float* const* t, const float* const* labels, const float* w, int k,
                                  const double* scratch, const float* gout, float* gs, float* const* gt, int batch, int channels,
                                  long long voxels, int bot, int top, float eps, void* stream);
/* nn.BCELoss() mean reduction (utils/evaluation.py:29-39), log clamped at -100 like torch */
int vs_bce_fwd(const float* p, const float* t, float* out, double* scratch, long long count, void* stream);
int vs_bce_bwd(const float* p, const float* t, const float* gout, float* gp, long long count, void* stream);

/* ---- optimiser / teacher ---------------------------------------------------------------------------- */
/* multi-tensor updates: ptr tables are DEVICE arrays of n_tensors device pointers, sizes[] element counts,
 * block_map[] = (tensor index, first element) pairs, one per 4096-element chunk (n_blocks of them).
 * SGD (torch.optim.SGD semantics, main_source.py:279-291): g += wd*p; buf = first ? g : mom*buf + g; p -= lr*buf */
int vs_sgd_momentum_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                          const int* block_map, int n_blocks, float lr, float momentum, float weight_decay,
                          int first_step, void* stream);
/* Adam (torch.optim.Adam, betas (b1,b2), eps 1e-8, L2 weight decay; main_source.py:292-294); step >= 1.  The betas are doubles
 * (python floats): 1 - beta and the bias corrections 1 - beta^step are formed in double, exactly as torch.optim.Adam forms them. */
int vs_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const long long* sizes, const int* block_map, int n_blocks, float lr, double beta1, double beta2,
                  float eps, float weight_decay, int step, void* stream);
/* ---- dynamic loss scaling: VS_F16 storage (BASELINE configs[4]; no counterpart in the reference, which trains in fp32) ------------
 * The soft-Dice gradients are O(1e-6): stored as fp16 they fall below the normal range.  The loss is therefore differentiated with
 * an upstream gradient of S = loss_scale[0] (a DEVICE scalar, so that a captured graph follows it), every gradient comes out S times
 * too large, and the optimiser divides by S.  Overflow protocol (torch.cuda.amp.GradScaler's, all on the device, no host sync):
 *   vs_grad_finite_multi      found_inf[0] = 1 if any gradient element is inf / nan (caller zero-initialises once)
 *   vs_*_scaled_multi         the update with g / S; the WHOLE step is skipped when found_inf[0] != 0  (NULL, NULL = the plain call)
 *   vs_loss_scale_update      found_inf ? S *= backoff, tracker = 0 : (++tracker == interval ? S *= growth, tracker = 0); found_inf = 0 */
int vs_sgd_momentum_scaled_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                                 const int* block_map, int n_blocks, float lr, float momentum, float weight_decay,
                                 int first_step, const float* loss_scale, const float* found_inf, void* stream);
/* the same update with the hyperparameters read from DEVICE memory, hyper[3] = {lr, momentum, weight_decay}: a launch captured into a HIP
 * graph (train.GraphedStep's captured tail) follows a learning-rate schedule without being re-captured — the host rewrites the three floats.
 * Momentum buffers must exist and hold zeros before the first step (buf = mom*0 + g = torch's clone(g)); loss_scale / found_inf may be NULL. */
int vs_sgd_momentum_dev_multi(float* const* params, const float* const* grads, float* const* bufs, const long long* sizes,
                              const int* block_map, int n_blocks, const float* hyper, const float* loss_scale,
                              const float* found_inf, void* stream);
int vs_adam_scaled_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                         const long long* sizes, const int* block_map, int n_blocks, float lr, double beta1, double beta2,
                         float eps, float weight_decay, int step, const float* loss_scale, const float* found_inf, void* stream);
int vs_grad_finite_multi(const float* const* grads, const long long* sizes, const int* block_map, int n_blocks,
                         float* found_inf, void* stream);
int vs_loss_scale_update(float* scale, int* growth_tracker, float* found_inf, float growth_factor, float backoff_factor,
                         int growth_interval, void* stream);
/* EMA teacher: t = alpha*t + (1-alpha)*s       (main_target.py:512-516) */
int vs_ema_multi(float* const* teacher, const float* const* student, const long long* sizes, const int* block_map,
                 int n_blocks, float alpha, void* stream);
/* dst_k[i] = src_k[i]*scale for every tensor k: gathers the (scattered) gradient tensors into the slices of one flat
 * all-reduce bucket, with the 1/world_size factor folded in — replaces nn.DataParallel's ReduceAddCoalesced staging
 * (main_source.py:354). */
int vs_copy_scale_multi(const float* const* srcs, float* const* dsts, const long long* sizes, const int* block_map,
                        int n_blocks, float scale, void* stream);
/* flat helper: dst[i] = src[i]*scale (dst may equal src: the 1/world scale after a sum all-reduce when the collective has no average) */
int vs_scale_copy(const float* src, float* dst, long long count, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAESEG_H */
