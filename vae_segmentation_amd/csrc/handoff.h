// The in-launch hand-off between workgroups of one SAMPLE (InstanceNorm3d's dependency domain, /root/reference/joint_model.py:11): arrival counters, bounded
// polling, and loads of statistics another workgroup of the same launch has just added to.  Used by the chains (chain.h, igemm_k3s.h) and by every
// backward-data kernel whose epilogue applies the InstanceNorm+ReLU backward to its own outputs (igemm_k3b.h / igemm_k3x.h EA, igemm.h g1_kernel).
// Protocol and its measurements: chain.h's header.
#pragma once
#include "common.h"

// all of this workgroup's stores (and atomics) of the phase are out; one lane signals for the workgroup
__device__ __forceinline__ void chain_arrive(unsigned int* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns once `target` workgroups have arrived (or the bounded spin gave up: fault word set)
__device__ __forceinline__ void chain_wait(unsigned int* ctr, unsigned int target, unsigned int* fault) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 18)) { atomicOr(fault, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// The same for large groups (a 24^3 layer: 144 workgroups per sample): the counter is kept in 8 shards on lines of their own — arrivals on one line retire
// ~12-20 ns apart (MI355X_MICROARCH.md fanin: 255 -> 1 in 3.2 us), so a single word costs a 144-workgroup group 2-3 us; a workgroup adds to shard
// (blockIdx.x & 7), lanes 0..7 of the polling wave read one shard each.
__device__ __forceinline__ void chain_arrive8(unsigned int* ctr8) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr8 + ((size_t)blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_wait8(unsigned int* ctr8, unsigned int target, unsigned int* fault) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int spins = 0;
        for (;;) {
            unsigned int v = lane < 8 ? __hip_atomic_load(ctr8 + (size_t)lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (__shfl(v, 0, 64) >= target) break;        // wave-uniform
            if (++spins > (1 << 18)) { if (lane == 0) atomicOr(fault, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// aux bits of the raw buffer intrinsics on gfx950: bit 0 = sc0, bit 1 = nt, bit 4 = sc1
#define VS_AUX_SC1 16

// stat_load() with sc1 loads: statistics another workgroup of this launch has just added to
__device__ __forceinline__ void stat_load_sc1(const double* st, size_t pair, size_t pairs, double (&out)[2]) {
#if VS_DET_BUILD
    long long a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s] = __double_as_longlong(__hip_atomic_load(st + stat_index(pair, pairs, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b[s] = __double_as_longlong(__hip_atomic_load(st + stat_index(pair, pairs, s) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    out[0] = ((double)a[0] * 0x1p40 + (double)a[1]) + ((double)a[2] * 0x1p-40 + (double)a[3] * 0x1p-80);
    out[1] = ((double)b[0] * 0x1p40 + (double)b[1]) + ((double)b[2] * 0x1p-40 + (double)b[3] * 0x1p-80);
#else
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int s = 0; s < VS_STAT_SLOTS; ++s) {
        a += __hip_atomic_load(st + stat_index(pair, pairs, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += __hip_atomic_load(st + stat_index(pair, pairs, s) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[0] = a; out[1] = b;
#endif
}
template <bool SC1>
__device__ __forceinline__ void stat_load_x(const double* st, size_t pair, size_t pairs, double (&out)[2]) {
    if constexpr (SC1) stat_load_sc1(st, pair, pairs, out);
    else stat_load(st, pair, pairs, out);
}

