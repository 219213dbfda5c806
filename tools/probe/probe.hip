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

// ---------------------------------------------------------------------------------------------------------------------------------------
// Composed Down head probe ("measure it"): nn.Conv3d(C, C, 2, stride 2) followed by nn.Conv3d(C, Co, 3, padding 1)
// with nothing in between (joint_model.py:126-136) as ONE linear operator on the fine grid — a 6x6x6 window with stride 2 and padding 2,
// Weff[pos] = W3[d] * W2[t] for pos = 2 d + t per axis, 216 taps of (Co x C) instead of 8 (C x C) + 27 (Co x C).  Forward only, bf16, C = 16, Co = 32,
// no lazy input (the product's kernels also normalise on load: this probe favours the composed form).  Direct-from-global like g1_kernel: a workgroup
// takes 256 output voxels (64 per wave, four column groups) and both 16-row blocks; k-group kg = taps (2 kg, 2 kg + 1) x 16 channels.
typedef __attribute__((ext_vector_type(4))) float pf32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int pu32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8;
__global__ __launch_bounds__(256) void down_composed_probe_kernel(const unsigned short* __restrict__ x, const pu32x4* __restrict__ wp, unsigned short* __restrict__ y,
                                                                   int N, int D, int H, int W) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, vcol = Do * Ho * Wo;
    const int tiles = (vcol + 255) / 256;
    const int n = blockIdx.x / tiles, tile = blockIdx.x - n * tiles;
    int oz[4], oy[4], ox[4];
    bool cvalid[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        int v = tile * 256 + wave * 64 + cg * 16 + col;
        cvalid[cg] = v < vcol;
        if (!cvalid[cg]) v = 0;
        ox[cg] = v % Wo; oy[cg] = (v / Wo) % Ho; oz[cg] = v / (Wo * Ho);
    }
    pf32x4 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[rb][cg] = pf32x4{0.f, 0.f, 0.f, 0.f};
    const int sub = g >> 1, ch0 = (g & 1) * 8;
    constexpr int NKG = 108, UN = 4;
    for (int kgb = 0; kgb < NKG; kgb += UN) {
        pu32x4 a[UN][2], b[UN][4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kg = kgb + u;
            a[u][0] = wp[(0 * NKG + kg) * 64 + lane];
            a[u][1] = wp[(1 * NKG + kg) * 64 + lane];
            const int tap = 2 * kg + sub;
            const int tz = tap / 36, ty = (tap / 6) % 6, tx = tap % 6;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const int iz = 2 * oz[cg] - 2 + tz, iy = 2 * oy[cg] - 2 + ty, ix = 2 * ox[cg] - 2 + tx;
                const bool ok = cvalid[cg] && (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                b[u][cg] = ok ? *(const pu32x4*)(x + ((((size_t)n * D + iz) * H + iy) * W + ix) * 16 + ch0) : pu32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    acc[rb][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8, a[u][rb]), __builtin_bit_cast(pbf16x8, b[u][cg]), acc[rb][cg], 0, 0, 0);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            if (!cvalid[cg]) continue;
            const size_t e = ((((size_t)n * Do + oz[cg]) * Ho + oy[cg]) * Wo + ox[cg]) * 32 + rb * 16 + 4 * g;
            typedef __attribute__((ext_vector_type(2))) float pf32x2;
            typedef __attribute__((ext_vector_type(2))) __bf16 pbf16x2;
            typedef __attribute__((ext_vector_type(2))) unsigned int pu32x2;
            pu32x2 pk;
            pk[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(pf32x2{acc[rb][cg][0], acc[rb][cg][1]}, pbf16x2));
            pk[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(pf32x2{acc[rb][cg][2], acc[rb][cg][3]}, pbf16x2));
            *(pu32x2*)(y + e) = pk;
        }
}
extern "C" int vs_debug_down_composed_probe(const void* x, const void* w_packed, void* y, int n, int d, int h, int w, void* stream) {
    if (!x || !w_packed || !y || n <= 0 || d <= 0 || h <= 0 || w <= 0 || ((d | h | w) & 1)) return -1;
    const int vcol = (d / 2) * (h / 2) * (w / 2);
    hipLaunchKernelGGL(down_composed_probe_kernel, dim3(((vcol + 255) / 256) * n), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (const pu32x4*)w_packed,
                       (unsigned short*)y, n, d, h, w);
    return (int)hipGetLastError();
}
