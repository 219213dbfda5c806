// Measurement aids — NOT part of the product ABI (include/vaeseg.h) and not linked into libvaeseg.so: tools/probe/libvsprobe.so, built by
// vae_segmentation_amd/csrc/Makefile, loaded by tools/probe/__init__.py (the tools/ scripts, and bench.py's live per-kernel timing pass for
// vs_spin).  No reference counterpart.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VS_OK 0
#define VS_EINVAL (-1)
#define VS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// ---- measurement aid: keep the queue busy for a given time (see include/vaeseg.h) -------------------------------------
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    for (int i = 0; i < (1 << 22); ++i) {                                     // bounded: every wave exits
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
extern "C" int vs_spin(int microseconds, void* stream) {
    if (microseconds < 0 || microseconds > 1000) return VS_EINVAL;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- measurement aid: cost of a device-wide barrier between resident workgroups ------------------------------------------
__global__ __launch_bounds__(256) void grid_barrier_probe_kernel(unsigned int* flags, unsigned long long* ticks, int n_wg, int iters, int mode) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    if ((mode & 0xff) == 2 || (mode & 0xff) == 3) {
        // Group protocol of an in-epilogue InstanceNorm-backward apply: groups of gsz = mode >> 8 workgroups; per iteration
        // every workgroup adds 32 fp64 partial sums (what the backward-data kernels do today), then — mode 2 only — signals its group's counter
        // (release), polls it, and reads the 32 totals back with device-scope atomic loads.  mode 3 = the atomics alone: the difference is the price.
        // ticks[n_wg ..] holds the sums: [group][parity][32] doubles (zeroed by the caller).
        const int gsz = mode >> 8, grp = blockIdx.x / gsz;
        double* sums = (double*)(ticks + n_wg) + (size_t)grp * 64;
        double sink = 0.0;
        for (int it = 1; it <= iters; ++it) {
            double* cur = sums + (it & 1) * 32;
            if (threadIdx.x < 32) atomicAdd(cur + threadIdx.x, 1.0);
            if ((mode & 0xff) == 2) {
                __syncthreads();
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(flags + grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    int spins = 0;
                    while (__hip_atomic_load(flags + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it * (unsigned int)gsz && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
                    if (spins >= (1 << 20)) s_fail = 1;
                }
                __syncthreads();
                if (s_fail) break;
                if (threadIdx.x < 32) sink += __hip_atomic_load(cur + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (threadIdx.x < 32 && sink == -1.0) flags[n_wg] = 1u;            // keep the loads alive
        if (threadIdx.x == 0) ticks[blockIdx.x] = s_fail ? ~0ull : __builtin_amdgcn_s_memrealtime() - t0;
        return;
    }
    if (mode == 1) {
        // one counter: every workgroup adds 1 and its thread 0 polls the counter (n serialized atomics + n pollers of one word)
        for (int it = 1; it <= iters; ++it) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(flags, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it * (unsigned int)n_wg && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(2);
                if (spins >= (1 << 20)) s_fail = 1;
            }
            __syncthreads();
            if (s_fail) break;
        }
        if (threadIdx.x == 0) ticks[blockIdx.x] = s_fail ? ~0ull : __builtin_amdgcn_s_memrealtime() - t0;
        return;
    }
    for (int it = 1; it <= iters; ++it) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned int)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < n_wg; i += 256) {
            int spins = 0;                               // bounded: a workgroup that is not resident must not hang the others
            while (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 20)) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) break;
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = s_fail ? ~0ull : __builtin_amdgcn_s_memrealtime() - t0;
}
extern "C" int vs_debug_grid_barrier_probe(unsigned int* flags, unsigned long long* ticks, int n_wg, int iters, int mode, void* stream) {
    if (!flags || !ticks || n_wg <= 0 || n_wg > 1024 || iters <= 0 || iters > 100000) return VS_EINVAL;
    hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, flags, ticks, n_wg, iters, mode);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// ---- measurement aid: what a kernel's stores add to the cost of a dependent graph node (tools/launch_floor.py) -------------------------
__global__ void store_probe_kernel(float* p, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) p[i] = 1.f;
    else if (mode == 1) __builtin_nontemporal_store(1.f, p + i);
    else if (mode == 2) __hip_atomic_store(p + i, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 3) __hip_atomic_store(p + i, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (mode == 4) { const float v = p[i]; if (v == 123.f) p[i + 1] = v; }       // load only
    else if (mode == 5) p[i] = p[i] + 1.f;                                              // read-modify-write: depends on the previous node's stores
    else if (mode == 6) p[i] = p[(i * 1031 + 7) % (gridDim.x * blockDim.x)] + 1.f;      // reads what OTHER workgroups (other XCDs) of the previous node wrote
}
extern "C" int vs_debug_store_probe(float* p, int n_wg, int mode, void* stream) {
    if (!p || n_wg <= 0 || n_wg > 65536) return VS_EINVAL;
    hipLaunchKernelGGL(store_probe_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, p, mode);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
