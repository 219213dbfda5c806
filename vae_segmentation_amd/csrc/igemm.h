// Implicit-GEMM convolution: shared parameter block and fragment conventions of every conv kernel, and g1_kernel — the
// direct-from-global kernel of the 2x2x2 stride-2 convolution (K2S2) and the pointwise + 2x scatter = transposed convolution (PW).
// The 3x3x3 kernels (LDS-staged halo tiles) are igemm_k3b.h / igemm_k3t.h / igemm_k3s.h (bf16, fp16) and igemm_k3.h (fp32).
//
// GEMM orientation: D[row = output channel m][col = voxel] = sum_k A[m][k] * B[k][voxel]
//   A = packed weights (fragment order, see pack.hip), 16 B/lane
//   B = activations: k runs over (tap, channel); a lane's 16-byte fragment is EPL contiguous channels of one input voxel
// The accumulator layout (col = lane&15, row = 4*(lane>>4)+reg) gives every lane 4 consecutive output
// channels of one voxel, i.e. one 16-byte (f32) / 8-byte (bf16, fp16) channels-last store.
#pragma once
#include <type_traits>
#include "common.h"
#include "handoff.h"

enum { G1_K3 = 0, G1_K2S2 = 1, G1_PW = 2 };
enum { EPI_RAW = 0, EPI_SOFTMAX2 = 1, EPI_SCATTER = 2 };

struct G1Params {
    const void* x;
    const double* x_stats;
    const void* wp;
    const float* bias;
    void* y;
    double* y_stats;
    float* prob;
    // optional fused InstanceNorm+ReLU-backward reduction on the OUTPUT g (backward-data use): with a = relu(norm(mx)),
    // sums[n][m] += (sum g*[xhat>0], sum g*[xhat>0]*xhat) over the voxels this kernel writes
    const void* mask_x;
    const double* mask_stats;
    double* sums;
    double inv_count_out;
    float drop_p;                   // SOFTMAX2 epilogue only: dropout on the two logits (0 = off)
    unsigned long long drop_seed;
    int N, D, H, W;       // input grid
    int Do, Ho, Wo;       // column grid (K3: = input; K2S2: input/2; PW: = input)
    int C;                // input channels (padded)
    int M;                // stored output channels (multiple of 8); for scatter: channels per tap
    int rb_total;         // 16-row blocks in the packed weight
    int nch;              // C / CK
    int tiles_per_sample; // column tiles per sample
    int tyn, txn;         // K3 tiling: tiles along y and x
    float eps;
    double inv_count_in;  // 1 / (D*H*W) of the input grid
    unsigned int fd_m[3], fd_s[3];   // k3b_kernel: multiply-shift pairs for / tiles_per_sample, / (txn*tyn), / txn (k3b_launch fills them)
    // k3t_kernel<FA>: backward-data with the IN-backward apply of its INPUT gradient fused into the staging: x = g (un-applied), x_stats = the
    // statistics of the activation fa_x, fa_sums its IN-backward sums, fa_dx (nullable) receives the applied gradient
    const void* fa_x;
    const double* fa_sums;
    void* fa_dx;
    // composed Up block (igemm_k4.h): tap lists per row block (forward) / per chunk (backward-data), bias table [27][Co], output channels of the 3x3x3 conv
    const void* up_taps;
    const float* up_btab;
    int up_co;
    // k3tw_kernel (igemm_k3tw.h): backward-data with the layer's weight gradient fused — one slab [27][8][8] per workgroup goes here
    float* wg_ws;
    // ... and its out_block form: the input gradient is the two-class softmax backward of (sm_prob, sm_gprob [+ sm_gcl]) formed while staging (drop_p / drop_seed
    // as in the forward); wg_bias (nullable) receives one (sum gl0, sum gl1) pair of doubles per workgroup
    const float* sm_prob;
    const float* sm_gprob;
    const void* sm_gcl;
    double* wg_bias;
    // k3b_kernel<..., EA> (igemm_k3b.h, round 6): backward-data whose epilogue also APPLIES the InstanceNorm+ReLU backward to its own outputs once the sample's
    // sums are complete — per-sample arrival counters ea_sync[n * 32] (zeroed by the caller), ea_items workgroups per sample, ea_fault: chain.h's fault word
    unsigned int* ea_sync;
    unsigned int* ea_fault;
    int ea_items;
    const void* ea_add;      // g1_kernel's epilogue apply only: a second gradient of the same raw tensor, summed in after the apply (vs_instnorm_relu_bwd_apply_add's form)
};

// LDS carve (bytes)
#define G1_LDS_MEAN 0       // float[256]
#define G1_LDS_RSTD 1024    // float[256]
#define G1_LDS_RED 2048     // float[4][64][2]
#define G1_LDS_BYTES 4096
#define G1_TILE_VOX 648     // 3x3x3 kernels: (4+2)*(4+2)*(16+2) halo voxels

template <typename T, int CK>
__device__ __forceinline__ u32x4 act_transform(u32x4 raw, const float* s_mean, const float* s_rstd, int c0) {
    constexpr int EPL = ET<T>::EPL;
    float v[EPL];
    frag_unpack(raw, v, (T*)nullptr);
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        float t = (v[j] - s_mean[c0 + j]) * s_rstd[c0 + j];
        v[j] = t > 0.f ? t : 0.f;
    }
    return frag_pack(v, (T*)nullptr);
}

// LIMB (fp32 storage only; round 5): the parity mode's stride-2 / transposed convolutions on the bf16 matrix cores.  The exact-f32 MFMA issues at 32 cycles per
// 16x16x4 product, and these launches are a few waves walking a long dependent MFMA chain (a 32-channel stride-2 chunk: 256 of them per wave and column set,
// 3.9 us) — they were the slowest launches per FLOP of the fp32 step.  Here TWO consecutive fp32 k-groups (2 x 4 k-values per lane, for A and B alike: any
// lane -> k assignment is valid as long as both operands share it) are split IN REGISTERS into three bf16 limbs each (common.h vs_limb_split4: the arithmetic of
// igemm_k3x.h) and multiplied by six v_mfma_f32_16x16x32_bf16 — 96 matrix cycles instead of 256 per (row block, column set), no new weight image: the packed fp32
// fragments are read as they are.  The leading product x0 w0 has its own accumulator where the register budget allows (RB <= 2), as in k3x_kernel.
// EA: the epilogue apply (see the epilogue) — instantiations of their own (16-row workgroups only): as a run-time branch its registers cost every launch of the
// family 0.3-0.5 us and the 64-row transposed conv at 96^3 16 us
template <typename T, int CK, int KIND, int MT, int EPI, bool LIMB = false, bool EA = false>
__global__ __launch_bounds__(256) void g1_kernel(const G1Params p) {
    using E = ET<T>;
    constexpr int EPL = E::EPL, KG = E::KG;
    static_assert(!LIMB || sizeof(T) == 4, "limb arithmetic: fp32 storage");
    static_assert(KIND == G1_K2S2 || KIND == G1_PW, "the 3x3x3 kernels live in igemm_k3b.h / igemm_k3.h");
    static_assert(EPI == EPI_RAW || EPI == EPI_SCATTER, "softmax epilogue: 3x3x3 kernels only");
    constexpr int NTAPS = KIND == G1_K2S2 ? 8 : 1;
    constexpr int NKG = (NTAPS * CK + KG - 1) / KG;
    constexpr int RB = MT / 16;
    constexpr int CKB = CK * (int)sizeof(T);          // bytes per voxel-chunk
    constexpr int TPK = KG > CK ? KG / CK : 1;        // taps covered by one k-group
    constexpr int KPT = KG > CK ? 1 : CK / KG;        // k-groups per tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_mean = (float*)(smem + G1_LDS_MEAN);
    float* s_rstd = (float*)(smem + G1_LDS_RSTD);
    float* s_red = (float*)(smem + G1_LDS_RED);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int col = lane & 15;
    const int g = lane >> 4;
    const int n = blockIdx.x / p.tiles_per_sample;
    const int tile = blockIdx.x - n * p.tiles_per_sample;
    const int rb0 = blockIdx.y * RB;
    const bool has_stats = p.x_stats != nullptr;
    const T* __restrict__ xin = (const T*)p.x;

    // ---- per-(n,c) mean / rstd of the lazy input ----
    if (has_stats) {
        for (int c = tid; c < p.C; c += 256) {
            float m, r;
            stats_to_mean_rstd(p.x_stats, (size_t)n * p.C + c, (size_t)p.N * p.C, p.inv_count_in, p.eps, m, r);
            s_mean[c] = m;
            s_rstd[c] = r;
        }
    }
    if (p.sums != nullptr) {
        // fused IN-backward sums (backward-data use; x_stats is NULL then, so the tables are free): mean / rstd of the
        // mask tensor's channels of this sample
        for (int c = tid; c < p.M; c += 256) {
            float m, r;
            stats_to_mean_rstd(p.mask_stats, (size_t)n * p.M + c, (size_t)p.N * p.M, p.inv_count_out, p.eps, m, r);
            s_mean[c] = m;
            s_rstd[c] = r;
        }
    }
    // ---- column geometry: 256 output voxels per workgroup, 64 per wave ----
    // Small volumes (<= 64 output voxels per sample: the 3^3 level, where only wave 0 would have real columns and the four waves would
    // each walk all C/32 chunks — one memory round trip per chunk — on padding): the waves share wave 0's columns and SPLIT the channel
    // chunks (wave w takes chunks w, w+4, ...); the partial accumulators are summed through LDS in wave order before the epilogue.
    // (The host picks this mode — 64-voxel column tiles, p.tyn == 64 — for every stride-2 conv with few workgroups and >= 2 chunks: four
    // times the workgroups, each walking a quarter of the chunks.)
    const bool splitw = KIND == G1_K2S2 && CK == 32 && p.tyn == 64;
    const int cwave = splitw ? 0 : wave;
    const int ctile = splitw ? 64 : 256;
    long long gofs[4];                          // element offset of the column's input voxel (tap 0)
    bool cvalid[4];
    int oz[4], oy[4], ox[4];
    {
        const int vcol = p.Do * p.Ho * p.Wo;             // < 2^31 (host check); 32-bit divisions: the 64-bit ones cost ~100 instructions each
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            int v = tile * ctile + cwave * 64 + cg * 16 + col;
            cvalid[cg] = v < vcol;
            if (!cvalid[cg]) v = 0;
            ox[cg] = v % p.Wo;
            const int t2 = v / p.Wo;
            oy[cg] = t2 % p.Ho;
            oz[cg] = t2 / p.Ho;
            const int s = KIND == G1_K2S2 ? 2 : 1;
            gofs[cg] = ((((long long)n * p.D + oz[cg] * s) * p.H + oy[cg] * s) * p.W + ox[cg] * s) * p.C;
        }
    }

    f32x4 acc[RB][4];
    constexpr bool TWO_ACC = LIMB && RB <= 2;            // the five small limb products apart from the leading one (rounded once per k-group pair, not six times)
    f32x4 acl[TWO_ACC ? RB : 1][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            acc[rb][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TWO_ACC) acl[TWO_ACC ? rb : 0][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // LIMB: fragments of two consecutive k-groups -> three bf16 limb fragments, six MFMAs (common.h vs_limb_pair / vs_limb_mfma)
    auto split_pair = [&](const u32x4& f0, const u32x4& f1, u32x4 (&out)[3]) { vs_limb_pair(f0, f1, out); };
    auto mfma_limbs = [&](const u32x4 (&al)[3], const u32x4 (&bl)[3], int rb, int cg) {
        vs_limb_mfma(al, bl, acc[rb][cg], TWO_ACC ? acl[TWO_ACC ? rb : 0][cg] : acc[rb][cg]);
    };
    const u32x4 zfrag = u32x4{0u, 0u, 0u, 0u};

    const u32x4* __restrict__ wp = (const u32x4*)p.wp;
    // The barrier that publishes the statistics tables is taken AFTER the first chunk's loads have been issued (each wave exactly once: in
    // its first chunk, or after the loop if it has none), so the statistics round trip and the first data round trip overlap instead of
    // following one another (2 dependent rounds -> 1 at the head of all 36 launches of this kernel per step).
    bool tables_published = false;
    auto publish_tables = [&]() {
        if (!tables_published) { __syncthreads(); tables_published = true; }
    };

    // The chunk loop is instantiated twice, with has_stats a compile-time constant: as a run-time flag its branch sat between
    // every direct-from-global B load and its use, and the compiler drained vmcnt(0) after each load (32 serialized memory
    // round trips per chunk).
    auto chunk_loop = [&](auto hs_tag) {
        constexpr bool HS = decltype(hs_tag)::value;
        for (int ch = splitw ? wave : 0; ch < p.nch; ch += splitw ? 4 : 1) {
            const u32x4* wch = wp + (size_t)ch * NKG * 64 + lane;
            const size_t rb_stride = (size_t)p.nch * NKG * 64;
            if constexpr (KG > CK) {
                // several taps per k-group: per-lane tap = kg*TPK + (g*EPL)/CK.  Every A and B fragment of the chunk (NKG <= 4 k-groups)
                // is requested before the first MFMA: issued inside the k-group loop each load was used at once and a 16-channel
                // stride-2 conv at 96^3 paid four memory round trips in a row.
                const int sub = (g * EPL) / CK;
                u32x4 aq[NKG][RB], bq[NKG][4];
#pragma unroll
                for (int kg = 0; kg < NKG; ++kg) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) aq[kg][rb] = wch[(size_t)(rb0 + rb) * rb_stride + kg * 64];
                    const int tap = kg * TPK + sub;
                    const int tt = tap < NTAPS ? tap : 0;
                    const int dz = (tt >> 2) & 1, dy = (tt >> 1) & 1, dx = tt & 1;
                    const long long toff = (((long long)dz * p.H + dy) * p.W + dx) * p.C + ch * CK + (g * EPL) % CK;
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) bq[kg][cg] = *(const u32x4*)(xin + gofs[cg] + toff);
                }
                __builtin_amdgcn_sched_barrier(0);
                publish_tables();
#pragma unroll
                for (int kg = 0; kg < NKG; ++kg) {
                    if (HS) {
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) bq[kg][cg] = act_transform<T, CK>(bq[kg][cg], s_mean, s_rstd, ch * CK + (g * EPL) % CK);
                    }
                    if constexpr (!LIMB) {
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                            for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = mfma16(aq[kg][rb], bq[kg][cg], acc[rb][cg], (T*)nullptr);
                    }
                }
                if constexpr (LIMB) {
#pragma unroll
                    for (int kg = 0; kg < NKG; kg += 2) {
                        u32x4 bl[4][3];
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) split_pair(bq[kg][cg], kg + 1 < NKG ? bq[kg + 1 < NKG ? kg + 1 : kg][cg] : zfrag, bl[cg]);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            u32x4 al[3];
                            split_pair(aq[kg][rb], kg + 1 < NKG ? aq[kg + 1 < NKG ? kg + 1 : kg][rb] : zfrag, al);
#pragma unroll
                            for (int cg = 0; cg < 4; ++cg) mfma_limbs(al, bl[cg], rb, cg);
                        }
                    }
                }
            } else {
                // wave-uniform tap; KPT k-groups per tap.
                constexpr int NK = NTAPS * KPT;
                if constexpr (NK * (4 + RB) <= 48) {
                    // Whole chunk in flight: every A and B fragment of the chunk is requested before the first is used, so a chunk
                    // costs one memory round trip (these layers have few workgroups — 2..100 — and nothing else hides the latency;
                    // the 32-deep chain of dependent loads made the 3^3 / 6^3 stride-2 convs the slowest launches per FLOP).
                    u32x4 aq[NK][RB], bq[NK][4];
#pragma unroll
                    for (int kg = 0; kg < NK; ++kg) {
                        const int tap = kg / KPT, kk = kg - tap * KPT;
                        const int dz = (tap >> 2) & 1, dy = (tap >> 1) & 1, dx = tap & 1;
                        const long long toff_g = (((long long)dz * p.H + dy) * p.W + dx) * p.C + ch * CK + kk * KG + g * EPL;
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) aq[kg][rb] = wch[(size_t)(rb0 + rb) * rb_stride + kg * 64];
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) bq[kg][cg] = *(const u32x4*)(xin + gofs[cg] + toff_g);
                    }
                    __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from sinking the requests back between the MFMAs
                    publish_tables();
#pragma unroll
                    for (int kg = 0; kg < NK; ++kg) {
                        const int cc = (kg % KPT) * KG + g * EPL;
                        if (HS) {
#pragma unroll
                            for (int cg = 0; cg < 4; ++cg) bq[kg][cg] = act_transform<T, CK>(bq[kg][cg], s_mean, s_rstd, ch * CK + cc);
                        }
                        if constexpr (!LIMB) {
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = mfma16(aq[kg][rb], bq[kg][cg], acc[rb][cg], (T*)nullptr);
                        }
                    }
                    if constexpr (LIMB) {
#pragma unroll
                        for (int kg = 0; kg < NK; kg += 2) {
                            u32x4 bl[4][3];
#pragma unroll
                            for (int cg = 0; cg < 4; ++cg) split_pair(bq[kg][cg], kg + 1 < NK ? bq[kg + 1 < NK ? kg + 1 : kg][cg] : zfrag, bl[cg]);
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb) {
                                u32x4 al[3];
                                split_pair(aq[kg][rb], kg + 1 < NK ? aq[kg + 1 < NK ? kg + 1 : kg][rb] : zfrag, al);
#pragma unroll
                                for (int cg = 0; cg < 4; ++cg) mfma_limbs(al, bl[cg], rb, cg);
                            }
                        }
                    }
                } else {
                // The A (weight) fragments come straight from global memory, so the
                // loop is software-pipelined: fragments for k-group kg+PD are requested while k-group kg is multiplied.
                constexpr int PD = LIMB ? (RB <= 2 ? 8 : 4) : (RB <= 2 ? 9 : 3);          // prefetch distance in k-groups (register budget RB*PD*4 VGPRs); LIMB: even (k-groups go in pairs)
                static_assert(!LIMB || NK % 2 == 0, "limb path: the pipelined chunk loop pairs k-groups");
                u32x4 abuf[PD][RB];
#pragma unroll
                for (int j = 0; j < PD; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        abuf[j][rb] = j < NK ? wch[(size_t)(rb0 + rb) * rb_stride + j * 64] : u32x4{0u, 0u, 0u, 0u};
                publish_tables();
#pragma unroll 1
                for (int kgb = 0; kgb < NK; kgb += PD) {
                    if constexpr (LIMB) {
#pragma unroll
                        for (int j = 0; j < PD; j += 2) {
                            const int kg = kgb + j;
                            if (kg < NK) {
                                u32x4 a2[2][RB], b2[2][4];
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                                    for (int rb = 0; rb < RB; ++rb) a2[h2][rb] = abuf[j + h2][rb];
                                    if (kg + h2 + PD < NK) {
#pragma unroll
                                        for (int rb = 0; rb < RB; ++rb) abuf[j + h2][rb] = wch[(size_t)(rb0 + rb) * rb_stride + (kg + h2 + PD) * 64];
                                    }
                                    const int tap = (kg + h2) / KPT, kk = (kg + h2) - tap * KPT;
                                    const int dz = (tap >> 2) & 1, dy = (tap >> 1) & 1, dx = tap & 1;
                                    const int cc = kk * KG + g * EPL;
                                    const long long toff_g = (((long long)dz * p.H + dy) * p.W + dx) * p.C + ch * CK + cc;
#pragma unroll
                                    for (int cg = 0; cg < 4; ++cg) b2[h2][cg] = *(const u32x4*)(xin + gofs[cg] + toff_g);
                                    if (HS) {
#pragma unroll
                                        for (int cg = 0; cg < 4; ++cg) b2[h2][cg] = act_transform<T, CK>(b2[h2][cg], s_mean, s_rstd, ch * CK + cc);
                                    }
                                }
                                u32x4 bl[4][3];
#pragma unroll
                                for (int cg = 0; cg < 4; ++cg) split_pair(b2[0][cg], b2[1][cg], bl[cg]);
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb) {
                                    u32x4 al[3];
                                    split_pair(a2[0][rb], a2[1][rb], al);
#pragma unroll
                                    for (int cg = 0; cg < 4; ++cg) mfma_limbs(al, bl[cg], rb, cg);
                                }
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < PD; ++j) {
                        const int kg = kgb + j;
                        if (kg < NK) {
                            u32x4 a[RB];
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb) a[rb] = abuf[j][rb];
                            if (kg + PD < NK) {
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb) abuf[j][rb] = wch[(size_t)(rb0 + rb) * rb_stride + (kg + PD) * 64];
                            }
                            const int tap = kg / KPT, kk = kg - tap * KPT;
                            u32x4 b[4];
                            const int dz = (tap >> 2) & 1, dy = (tap >> 1) & 1, dx = tap & 1;
                            const int cc = kk * KG + g * EPL;
                            const long long toff_g = (((long long)dz * p.H + dy) * p.W + dx) * p.C + ch * CK + cc;
#pragma unroll
                            for (int cg = 0; cg < 4; ++cg) b[cg] = *(const u32x4*)(xin + gofs[cg] + toff_g);
                            if (HS) {
#pragma unroll
                                for (int cg = 0; cg < 4; ++cg) b[cg] = act_transform<T, CK>(b[cg], s_mean, s_rstd, ch * CK + cc);
                            }
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = mfma16(a[rb], b[cg], acc[rb][cg], (T*)nullptr);
                        }
                    }
                }
                }
            }
        }
    };
    if (has_stats) chunk_loop(std::true_type{}); else chunk_loop(std::false_type{});
    if constexpr (TWO_ACC) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[rb][cg] += acl[TWO_ACC ? rb : 0][cg];
    }
    publish_tables();
    if (splitw) {
        // fixed order w0 + w1 + w2 + w3 (bitwise reproducible); waves 1-3 then hold no columns of their own
        f32x4* s_part = (f32x4*)(smem + G1_LDS_BYTES);   // [3 waves][RB][4][64 lanes]
        if (wave > 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) s_part[(((wave - 1) * RB + rb) * 4 + cg) * 64 + lane] = acc[rb][cg];
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) {
                        const f32x4 o = s_part[((w * RB + rb) * 4 + cg) * 64 + lane];
                        acc[rb][cg][0] += o[0]; acc[rb][cg][1] += o[1]; acc[rb][cg][2] += o[2]; acc[rb][cg][3] += o[3];
                    }
        } else {
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) cvalid[cg] = false;
        }
    }

    // ------------------------------------------------------------------------------------------
    // epilogues
    // ------------------------------------------------------------------------------------------
    T* __restrict__ yout = (T*)p.y;
    float ssum[RB][4], ssq[RB][4], mk_mean[RB][4], mk_rstd[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[rb][r] = 0.f; ssq[rb][r] = 0.f; mk_mean[rb][r] = 0.f; mk_rstd[rb][r] = 1.f; }

    // element offset of every (row block, column group) fragment this lane stores
    auto out_elem = [&](int rb, int cg, int& m_out, bool& valid) -> size_t {
        const int row = (rb0 + rb) * 16 + 4 * g;          // first of this lane's 4 consecutive rows
        int m = row, tap = 0;
        if constexpr (EPI == EPI_SCATTER) { tap = row / p.M; m = row - tap * p.M; }
        const bool rvalid = EPI == EPI_SCATTER ? (tap < 8) : (row < p.M);
        m_out = m;
        valid = rvalid && cvalid[cg];
        if constexpr (EPI == EPI_SCATTER) {
            const int dz = (tap >> 2) & 1, dy = (tap >> 1) & 1, dx = tap & 1;
            return ((((size_t)n * (2 * p.D) + 2 * oz[cg] + dz) * (2 * p.H) + 2 * oy[cg] + dy) * (2 * p.W) + 2 * ox[cg] + dx) * p.M + m;
        } else {
            return ((((size_t)n * p.Do + oz[cg]) * p.Ho + oy[cg]) * p.Wo + ox[cg]) * p.M + m;
        }
    };
    // backward-data use: ALL mask fragments are requested before the first store (a load issued between the stores of the
    // loop below cannot be hoisted over them by the compiler — the tensors may alias for all it knows — and each would cost a
    // full memory round trip: 16 in a row for a 64-row scatter workgroup)
    constexpr int MKW = sizeof(T) == 4 ? 4 : 2;
    unsigned int mkv[RB][4][MKW];
    // Epilogue apply (round 6; backward-data use with fused sums, every workgroup of the launch resident — the stride-2 / transposed launches of the <= 48^3 levels):
    // the workgroup keeps its outputs and mask values in registers, adds its partial sums, arrives on its SAMPLE's counter, waits for the sample's other workgroups,
    // reads the complete sums back and stores the APPLIED gradient rstd * (g * [xhat > 0] - m1 - xhat * m2) [rounded, + ea_add] — the arithmetic of
    // vs_instnorm_relu_bwd_apply_add on the same rounded values; the un-applied tensor is never written and the standalone apply launch (~5 us) disappears.
    constexpr bool ea = EA;
    static_assert(!EA || MT == 16, "epilogue apply: 16-row workgroups");
    unsigned int adv[EA ? RB : 1][4][MKW];
    if (p.sums != nullptr) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                int m; bool ok;
                const size_t e = out_elem(rb, cg, m, ok);
#pragma unroll
                for (int i = 0; i < MKW; ++i) { mkv[rb][cg][i] = 0u; if constexpr (EA) adv[rb][cg][i] = 0u; }
                if (ok) {
                    if constexpr (sizeof(T) == 4) {
                        const u32x4 xx = *(const u32x4*)((const float*)p.mask_x + e);
                        mkv[rb][cg][0] = xx[0]; mkv[rb][cg][1] = xx[1]; mkv[rb][cg][2] = xx[2]; mkv[rb][cg][3] = xx[3];
                        if constexpr (EA) {
                            if (p.ea_add != nullptr) {
                                const u32x4 aa = *(const u32x4*)((const float*)p.ea_add + e);
                                adv[rb][cg][0] = aa[0]; adv[rb][cg][1] = aa[1]; adv[rb][cg][2] = aa[2]; adv[rb][cg][3] = aa[3];
                            }
                        }
                    } else {
                        const u32x2 xx = *(const u32x2*)((const T*)p.mask_x + e);
                        mkv[rb][cg][0] = xx[0]; mkv[rb][cg][1] = xx[1];
                        if constexpr (EA) {
                            if (p.ea_add != nullptr) {
                                const u32x2 aa = *(const u32x2*)((const T*)p.ea_add + e);
                                adv[rb][cg][0] = aa[0]; adv[rb][cg][1] = aa[1];
                            }
                        }
                    }
                }
            }
    }

    float bvq[RB][4];                            // bias rows, requested up front for the same reason
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        int m; bool ok0;
        (void)out_elem(rb, 0, m, ok0);
        const int row = (rb0 + rb) * 16 + 4 * g;
        const bool rvalid = EPI == EPI_SCATTER ? (row / p.M < 8) : (row < p.M);
#pragma unroll
        for (int r = 0; r < 4; ++r) bvq[rb][r] = (p.bias && rvalid) ? p.bias[m + r] : 0.f;
    }

#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        int m; bool ok0;
        (void)out_elem(rb, 0, m, ok0);
        const int row = (rb0 + rb) * 16 + 4 * g;
        const bool rvalid = EPI == EPI_SCATTER ? (row / p.M < 8) : (row < p.M);
        float bv[4] = {bvq[rb][0], bvq[rb][1], bvq[rb][2], bvq[rb][3]};
        if (p.sums != nullptr && rvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { mk_mean[rb][r] = s_mean[m + r]; mk_rstd[rb][r] = s_rstd[m + r]; }
        }
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            int m2; bool ok;
            const size_t e = out_elem(rb, cg, m2, ok);
            if (!ok) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = E::rnd(acc[rb][cg][r] + bv[r]);
            if (!ea) store4<T>(yout + e, v);
            if (p.sums != nullptr) {
                float xv[4];
                widen4<T>(mkv[rb][cg], xv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (xv[r] - mk_mean[rb][r]) * mk_rstd[rb][r];
                    const float gm = xh > 0.f ? v[r] : 0.f;
                    ssum[rb][r] += gm; ssq[rb][r] += gm * xh;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[rb][r] += v[r]; ssq[rb][r] += v[r] * v[r]; }
            }
        }
    }

    double* const red_dst0 = p.sums != nullptr ? p.sums : p.y_stats;
    double* const red_dst = red_dst0;
    if ((EPI == EPI_RAW || EPI == EPI_SCATTER) && red_dst != nullptr) {
        // reduce over the 16 columns held by lanes with equal g, then over waves, then one fp64 atomic per (m, stat)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[rb][r], q = ssq[rb][r];
                { s = row16_sum(s); q = row16_sum(q); }      // DPP: the four-step __shfl_xor butterfly was four ds_bpermute round trips per statistic
                if (col == 0) {
                    const int lr = rb * 16 + 4 * g + r;      // row within the WG's MT rows
                    s_red[(wave * 64 + lr) * 2 + 0] = s;
                    s_red[(wave * 64 + lr) * 2 + 1] = q;
                }
            }
        __syncthreads();
        // one atomic per (channel, statistic) per workgroup; scatter rows (tap, channel) of equal channel are folded first
        const int nch_here = (EPI == EPI_SCATTER && p.M < MT) ? p.M : MT;
        if (tid < nch_here * 2) {
            const int lc = tid >> 1, st = tid & 1;
            double tot = 0.0;
            int ch_out = -1;
            for (int lr = lc; lr < MT; lr += nch_here) {
                const int row = rb0 * 16 + lr;
                const bool rok = EPI == EPI_SCATTER ? (row < 8 * p.M) : (row < p.M);
                if (rok) {
                    ch_out = EPI == EPI_SCATTER ? row % p.M : row;
                    tot += (double)s_red[(0 * 64 + lr) * 2 + st] + (double)s_red[(1 * 64 + lr) * 2 + st] +
                           (double)s_red[(2 * 64 + lr) * 2 + st] + (double)s_red[(3 * 64 + lr) * 2 + st];
                }
            }
            if (ch_out >= 0) stat_add(red_dst, (size_t)n * p.M + ch_out, (size_t)p.N * p.M, st, tot);
        }
    }
    if constexpr (EA) {
        // ---- the sample's sums are complete once all of its workgroups have arrived; then the apply, on registers (k3b_kernel<..., EA>'s form) ----
        unsigned int* ctr = p.ea_sync + (size_t)n * 256;     // 8 shards of 128 bytes per sample
        chain_arrive8(ctr);
        chain_wait8(ctr, (unsigned int)p.ea_items, p.ea_fault);
        if (tid < MT) {
            const int row = rb0 * 16 + tid;
            const bool rok = EPI == EPI_SCATTER ? (row < 8 * p.M) : (row < p.M);
            const int ch = EPI == EPI_SCATTER ? row % p.M : row;
            float m = 0.f, r = 1.f, a = 0.f, b = 0.f;
            if (rok) {
                stats_to_mean_rstd(p.mask_stats, (size_t)n * p.M + ch, (size_t)p.N * p.M, p.inv_count_out, p.eps, m, r);     // the standalone apply's exact form
                double sv[2];
                stat_load_sc1(p.sums, (size_t)n * p.M + ch, (size_t)p.N * p.M, sv);
                a = (float)(sv[0] * p.inv_count_out);
                b = (float)(sv[1] * p.inv_count_out);
            }
            *(f32x4*)(s_red + tid * 4) = f32x4{m, r, a, b};
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x4 tb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[r] = *(const f32x4*)(s_red + (rb * 16 + 4 * g + r) * 4);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                int m2; bool ok;
                const size_t e = out_elem(rb, cg, m2, ok);
                if (!ok) continue;
                float xv[4], av[4], o[4];
                widen4<T>(mkv[rb][cg], xv);
                widen4<T>(adv[rb][cg], av);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gv = E::rnd(acc[rb][cg][r] + bvq[rb][r]);
                    o[r] = vs_in_bwd_apply1(gv, xv[r], tb[r][0], tb[r][1], tb[r][2], tb[r][3]);
                    if (p.ea_add != nullptr) o[r] = E::rnd(o[r]) + av[r];
                }
                store4<T>(yout + e, o);
            }
        }
    }
}

template <typename T, int CK, int KIND, int MT, int EPI, bool LIMB = false, bool EA = false>
static int g1_launch(const G1Params& p, int tiles_total, int row_tiles, hipStream_t stream) {
    constexpr size_t lds = G1_LDS_BYTES + (KIND == G1_K2S2 && CK == 32 ? (size_t)3 * (MT / 16) * 4 * 64 * 16 : 0);   // + the wave-split partials
    auto kern = g1_kernel<T, CK, KIND, MT, EPI, LIMB, EA>;
    hipLaunchKernelGGL(kern, dim3(tiles_total, row_tiles), dim3(256), lds, stream, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
