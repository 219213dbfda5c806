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

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6, VERDICT r05 item 1 step 0: the price of ONE layer boundary kept inside a launch when the dependency domain is a SAMPLE
// (InstanceNorm3d is per (n, c): joint_model.py:11), not the device.  Groups of gsz workgroups stand for the workgroups of one sample's layer.
// Per round every workgroup: stores `pb` bytes of payload (its output tile), publishes 32 fp64 partial sums (its statistics), drains
// (`s_waitcnt vmcnt(0)`), workgroup barrier, ONE relaxed agent-scope add to the group's counter, ONE lane polls the counter with sc1 loads;
// workgroup barrier; then the "next layer's staging": the neighbour's payload and all gsz partial slots are loaded and checked (a stale
// payload word — value != round — is counted).  No release / acquire fence anywhere in the fast forms (no buffer_wbl2 / buffer_inv).
//   mode bit 0: members of a group share blockIdx % 8 (one XCD under round-robin placement) instead of being consecutive (spread over XCDs)
//   mode bit 1: payload stores plain (kept in the XCD's L2) instead of sc1 (write-through)
//   mode bit 2: payload / partial loads sc0 instead of sc1 (the review's proposal; MI355X_MICROARCH.md says sc0 loads hit L1 like plain)
//   mode bit 3: statistics by fp64 agent atomics into one table per group instead of one slot per workgroup
//   mode bit 4: the fenced protocol for comparison: plain stores, release fence before the add, acquire fence after the poll, plain loads
// out: ticks[b] = memrealtime ticks (100 MHz) of the whole loop or ~0 on a bounded-spin give-up; xcc[b] = HW_REG_XCC_ID; err[0] += stale words.
typedef __attribute__((ext_vector_type(4))) unsigned int qu32x4;
__device__ __forceinline__ void st16_sc1(void* p, qu32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ qu32x4 ld16_sc1(const void* p) {
    qu32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ qu32x4 ld16_sc0(const void* p) {
    qu32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(256) void xcd_group_probe_kernel(unsigned int* flags, unsigned long long* ticks, unsigned int* xcc, unsigned int* err,
                                                               unsigned char* payload, double* slots, int n_wg, int iters, int gsz, int pb, int mode) {
    __shared__ int s_fail;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) {
        s_fail = 0;
        xcc[b] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;      // HW_REG_XCC_ID (id 20), bits [3:0]
    }
    int grp, mem;
    if (mode & 1) { const int x = b & 7, j = b >> 3, per = (n_wg >> 3) / gsz; grp = x * per + j / gsz; mem = j % gsz; if (j / gsz >= per) return; }
    else { grp = b / gsz; mem = b % gsz; }
    const int base = (mode & 1) ? -1 : grp * gsz;
    auto member_block = [&](int m) { return (mode & 1) ? ((((grp % ((n_wg >> 3) / gsz)) * gsz + m) << 3) | (grp / ((n_wg >> 3) / gsz))) : base + m; };
    const bool plain_st = mode & 2, sc0_ld = mode & 4, atom = mode & 8, fenced = mode & 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double sink = 0.0;
    unsigned int stale = 0;
    const int nb = member_block((mem + 1) % gsz);
    for (int it = 1; it <= iters; ++it) {
        const int par = it & 1;
        unsigned char* mine = payload + ((size_t)par * n_wg + b) * pb;
        const qu32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
        for (int o = tid * 16; o < pb; o += 4096) {
            if (plain_st || fenced) *(qu32x4*)(mine + o) = v;
            else st16_sc1(mine + o, v);
        }
        if (tid < 32) {
            if (atom) atomicAdd(slots + ((size_t)par * n_wg + grp) * 32 + tid, 1.0);
            else if (fenced) slots[((size_t)par * n_wg + b) * 32 + tid] = (double)it;
            else __hip_atomic_store(slots + ((size_t)par * n_wg + b) * 32 + tid, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(flags + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(flags + grp * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it * (unsigned int)gsz && ++spins < (1 << 18)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 18)) s_fail = 1;
            if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        if (s_fail) break;
        const unsigned char* theirs = payload + ((size_t)par * n_wg + nb) * pb;
        for (int o = tid * 16; o < pb; o += 4096) {
            const qu32x4 r = fenced ? *(const qu32x4*)(theirs + o) : sc0_ld ? ld16_sc0(theirs + o) : ld16_sc1(theirs + o);
            stale += (r[0] != (unsigned)it) + (r[3] != (unsigned)it);
        }
        if (tid < 32) {
            if (atom) sink += __hip_atomic_load(slots + ((size_t)par * n_wg + grp) * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                for (int m = 0; m < gsz; ++m) {
                    const double* q = slots + ((size_t)par * n_wg + member_block(m)) * 32 + tid;
                    const double s = fenced ? *q : __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stale += (s != (double)it);
                    sink += s;
                }
        }
    }
    if (stale) atomicAdd(err, stale);
    if (tid < 32 && sink == -1.0) err[1] = 1u;
    if (tid == 0) ticks[b] = s_fail ? ~0ull : __builtin_amdgcn_s_memrealtime() - t0;
}
extern "C" int vs_debug_xcd_group_probe(unsigned int* flags, unsigned long long* ticks, unsigned int* xcc, unsigned int* err, void* payload, double* slots,
                                        int n_wg, int iters, int gsz, int pb, int mode, void* stream) {
    if (!flags || !ticks || !xcc || !err || !payload || !slots || n_wg <= 0 || n_wg > 256 || (n_wg & 7) || iters <= 0 || iters > 100000) return VS_EINVAL;
    if (gsz <= 0 || pb < 4096 || (pb & 4095) || pb > (64 << 10)) return VS_EINVAL;
    if ((mode & 1) ? ((n_wg >> 3) % gsz != 0) : (n_wg % gsz != 0)) return VS_EINVAL;      // every workgroup belongs to a full group: nobody waits for an absent member
    hipLaunchKernelGGL(xcd_group_probe_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, flags, ticks, xcc, err, (unsigned char*)payload, slots, n_wg, iters, gsz, pb, mode);
    VS_CHECK_LAUNCH();
    return VS_OK;
}

// the same round as a chain of dependent launches (what the step does today): launch k stores its payload + partials, launch k + 1 reads them
__global__ __launch_bounds__(256) void xcd_chain_probe_kernel(unsigned int* err, unsigned char* payload, double* slots, int n_wg, int gsz, int pb, int it) {
    const int tid = threadIdx.x, b = blockIdx.x, grp = b / gsz, mem = b % gsz, par = it & 1, prev = par ^ 1;
    unsigned int stale = 0;
    double sink = 0.0;
    if (it > 1) {
        const unsigned char* theirs = payload + ((size_t)prev * n_wg + grp * gsz + (mem + 1) % gsz) * pb;
        for (int o = tid * 16; o < pb; o += 4096) { const qu32x4 r = *(const qu32x4*)(theirs + o); stale += (r[0] != (unsigned)(it - 1)); }
        if (tid < 32) for (int m = 0; m < gsz; ++m) { const double s = slots[((size_t)prev * n_wg + grp * gsz + m) * 32 + tid]; stale += (s != (double)(it - 1)); sink += s; }
    }
    unsigned char* mine = payload + ((size_t)par * n_wg + b) * pb;
    const qu32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
    for (int o = tid * 16; o < pb; o += 4096) *(qu32x4*)(mine + o) = v;
    if (tid < 32) slots[((size_t)par * n_wg + b) * 32 + tid] = (double)it + (sink == -1.0 ? 1.0 : 0.0);
    if (stale) atomicAdd(err, stale);
}
extern "C" int vs_debug_xcd_chain_probe(unsigned int* err, void* payload, double* slots, int n_wg, int gsz, int pb, int it, void* stream) {
    if (!err || !payload || !slots || n_wg <= 0 || n_wg > 1024 || gsz <= 0 || n_wg % gsz || pb < 4096 || (pb & 4095) || it < 1) return VS_EINVAL;
    hipLaunchKernelGGL(xcd_chain_probe_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, err, (unsigned char*)payload, slots, n_wg, gsz, pb, it);
    VS_CHECK_LAUNCH();
    return VS_OK;
}
