// Several dependent layers of ONE sample inside one launch (round 6).
//
// The levels at 12^3 and below hold < 10 % of the step's bytes and ~ 140 of its 206 launches: every launch there is a few microseconds of
// work inside ~ 9 us of launch boundary, prologue and cold tables (profiles/r04_k3b_phase_stamps.txt).  InstanceNorm3d makes the dependency
// domain of a layer ONE SAMPLE (joint_model.py:11: per-(n, c) statistics, no affine, no running state), not the device: the workgroups of
// sample n's layer l + 1 need sample n's layer l complete — its raw output and its (sum, sumsq) — and nothing of any other sample.
// So a chain kernel keeps a DoubleConv's three convolutions (forward), or its three backward-data convolutions and the InstanceNorm+ReLU
// backward applies between them (backward), in one launch: with K = c.xcds XCD slots per sample, workgroup b (slot = b & 7, j = b >> 3) is item
// j * K + slot % K of the samples slot / K, slot / K + 8 / K, ...; under the dispatcher's round-robin placement the workgroups of one sample share
// one XCD (K = 1: at most 32 of them, one per CU — the chain kernels run one workgroup per CU and every workgroup of a sample must be resident
// while its peers wait for it) or K neighbouring ones.  Placement is for speed and residency only: the hand-off below is valid for any placement.
//
// Hand-off between layers (MI355X_MICROARCH.md, Workgroup dispatch ... inter-workgroup visibility, "Valid forms", first table row; measured
// for this pattern by tools/xcd_probe.py -> profiles/r06_xcd_barrier_probe.txt): every byte handed over is stored write-through (sc1) and
// loaded with sc1 loads (L1 bypassed), the statistics travel by agent-scope atomics as in every other kernel of the library; every storing
// wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds 1 to the (sample, phase) counter
// (relaxed, agent scope); a consumer's lane 0 polls that counter with sc1 loads until all `items` workgroups of the sample have arrived,
// then the workgroup barrier releases the other waves.  No release / acquire fence (no buffer_wbl2 / buffer_inv): 1.8-3.0 us per boundary
// for 8-32 workgroups against 4.3-5.4 us for a dependent launch.
// Every spin is bounded: a workgroup that waits for one that is not resident (a grid that does not fit the chip: rejected on the host) gives
// up after ~0.2 s, raises the fault word the host reads back (ops.chain_fault) and runs on — wrong numbers, never a hung queue.
#pragma once
#include "igemm.h"

#ifndef K3S_NCG
#define K3S_NCG 2                    // igemm_k3s.h: 16-voxel column groups per workgroup of the 6^3-class volumes (the chain planner needs the tile width too)
#endif
#ifndef K3S_NCG_SMALL
#define K3S_NCG_SMALL 1              // ... and of the 3^3-class volumes (27 voxels: one group per workgroup, two workgroups — fp32 mode 5.735 -> 5.692 ms same box; two groups in one: 2)
#endif
#define VS_CHAIN_MAX_LAYERS 3
#define VS_CHAIN_PHASES 8            // counters per sample, one 128-byte line each
#define VS_CHAIN_MAX_ITEMS 256       // workgroups per sample
#define VS_CHAIN_XCD_ITEMS 32        // ... of which at most one per CU of an XCD (the chain kernels run one workgroup per CU: every workgroup of a sample must be resident)

struct K3Chain {
    G1Params p[VS_CHAIN_MAX_LAYERS]; // in execution order
    int nl;
    int items;                       // workgroups per sample (the same for every layer of the chain)
    int xcds;                        // XCD slots a sample's workgroups are dealt over: 1, 2, 4 or 8 (items <= 32 * xcds)
    int apply_mask;                  // backward chains: bit l = layer l's output gets the InstanceNorm+ReLU backward apply in place before the next layer reads it
    unsigned int* sync;              // zeroed by the caller: [N][VS_CHAIN_PHASES][32]
    unsigned int* fault;             // device word, never cleared by the kernels
    const void* add;                 // backward chains: a second gradient of the LAST layer's activation summed in by its apply (ops._park_gradient)
};

// this workgroup's place in the chain: first sample, sample stride, item (>= c.items: none — the grid is rounded up to whole XCD rounds)
struct ChainPlace { int n0, nstep, item; };
__device__ __forceinline__ ChainPlace chain_place(const K3Chain& c) {
    const int slot = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, k = c.xcds;
    ChainPlace pl;
    pl.n0 = slot / k; pl.nstep = 8 / k; pl.item = j * k + slot % k;
    return pl;
}
static inline int chain_grid(K3Chain& c) {               // host: choose xcds, -> workgroups to launch (0: the chain does not fit)
    if (c.items <= 0 || c.items > VS_CHAIN_MAX_ITEMS) return 0;
    int k = 1;
    while (c.items > VS_CHAIN_XCD_ITEMS * k) k *= 2;
    c.xcds = k;
    return 8 * ((c.items + k - 1) / k);
}

__device__ __forceinline__ unsigned int* chain_counter(const K3Chain& c, int n, int phase) { return c.sync + ((size_t)n * VS_CHAIN_PHASES + phase) * 32; }

// The InstanceNorm+ReLU backward apply of one sample, in place, by the `items` workgroups of the sample (norm.hip in_relu_bwd_apply_body's
// arithmetic, value for value): g <- rstd * (g * [xhat > 0] - m1 - xhat * m2) [then rounded + add].  g and the sums were produced inside this
// launch (sc1 loads); x and its statistics by an earlier one.  s_tab: 4 * c floats of LDS.
template <typename T>
__device__ __forceinline__ void chain_apply(void* g_, const void* x_, const double* xs, const double* sums, const void* add_, int n, int nn, int voxels, int c,
                                            double inv_count, float eps, int item, int items, float* s_tab) {
    constexpr int EPL = ET<T>::EPL, ES = (int)sizeof(T);
    float *s_m = s_tab, *s_r = s_tab + c, *s_a = s_tab + 2 * c, *s_b = s_tab + 3 * c;
    const int nt = (int)blockDim.x;
    for (int i = threadIdx.x; i < c; i += nt) {
        float m, r;
        stats_to_mean_rstd(xs, (size_t)n * c + i, (size_t)nn * c, inv_count, eps, m, r);
        double sv[2];
        stat_load_sc1(sums, (size_t)n * c + i, (size_t)nn * c, sv);
        s_m[i] = m; s_r[i] = r;
        s_a[i] = (float)(sv[0] * inv_count);
        s_b[i] = (float)(sv[1] * inv_count);
    }
    __syncthreads();
    const int frags = c / EPL, total = voxels * frags;
    const unsigned int bytes = (unsigned int)((long long)nn * voxels * c * ES);
    const i32x4 grsrc = make_rsrc(g_, bytes), xrsrc = make_rsrc(x_, bytes), arsrc = make_rsrc(add_ ? add_ : g_, bytes);
    const int sample = n * voxels * c * ES;
    for (int f = item * nt + (int)threadIdx.x; f < total; f += items * nt) {
        const int fx = f % frags;
        const int off = sample + f * 16;
        const u32x4 gq = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(grsrc, off, 0, VS_AUX_SC1));
        const u32x4 xq = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(xrsrc, off, 0, 0));
        float fg[EPL], fv[EPL], o[EPL];
        frag_unpack(gq, fg, (T*)nullptr);
        frag_unpack(xq, fv, (T*)nullptr);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int ch = fx * EPL + j;
            o[j] = vs_in_bwd_apply1(fg[j], fv[j], s_m[ch], s_r[ch], s_a[ch], s_b[ch]);
        }
        if (add_ != nullptr) {
            const u32x4 aq = __builtin_bit_cast(u32x4, vs_raw_buffer_load_b128(arsrc, off, 0, 0));
            float fa[EPL];
            frag_unpack(aq, fa, (T*)nullptr);
#pragma unroll
            for (int j = 0; j < EPL; ++j) o[j] = ET<T>::rnd(o[j]) + fa[j];
        }
        vs_raw_buffer_store_b128(__builtin_bit_cast(i32x4, frag_pack(o, (T*)nullptr)), grsrc, off, 0, VS_AUX_SC1);
    }
}
