"""GPU-box aid: the weight gradient of an 8 -> 8 3x3x3 layer at full resolution FUSED into its backward-data launch
(vs_conv_k3_bwd_data_wgrad, csrc/igemm_k3tw.h) against what the library ran for it before: the backward-data launch + the layer's share of the grouped
weight-gradient launch.  One layer in isolation, back-to-back launches on one stream, with and without the fused apply of the incoming gradient.
usage: python tools/k3tw_probe.py [N S iters [fp16]]      (record: profiles/r05_k3tw_probe.txt)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_segmentation_amd import ops                                              # noqa: E402
from vae_segmentation_amd._lib import VS_CONV_K3, check, lib                      # noqa: E402

C = 8


def run(n, s, iters, dt):
    d = h = w = s
    gen = torch.Generator().manual_seed(0)
    ax = (torch.randn(n, d, h, w, C, generator=gen) * 1.3 + 0.2).to(dt).cuda()      # raw output of the layer whose gradient arrives un-applied
    g = torch.randn(n, d, h, w, C, generator=gen).to(dt).cuda()                     # dL/d relu(norm(ax))
    mx = (torch.randn(n, d, h, w, C, generator=gen) * 0.8 - 0.1).to(dt).cuda()      # raw input of the conv (lazy activation)
    wt = (torch.randn(C, C, 3, 3, 3, generator=gen) * 0.1).cuda()
    ops.stats_arena_begin(ax.device)
    axs, mxs = ops.instnorm_stats(ax), ops.instnorm_stats(mx)
    vox, vdt, st = d * h * w, ops.vs_dtype(ax), ops._stream()
    asums = ops._new_stats(n, C, ax.device)
    check(lib.vs_instnorm_relu_bwd_reduce(g.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), n, vox, C, vdt, 1e-5, st), "reduce")
    wpb = ops.pack_weight(wt, ops.VS_PACK_ROWS_D1_FLIP, C, dt)
    dx_ref = torch.empty_like(g)
    check(lib.vs_instnorm_relu_bwd_apply(g.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), dx_ref.data_ptr(), n, vox, C, vdt, 1e-5, st), "apply")
    y = torch.empty_like(mx)
    sums = ops._new_stats(n, C, ax.device)
    dx = torch.empty_like(g)
    nslabs = lib.vs_conv_k3_bwd_data_wgrad_slabs(n, d, h, w)
    slabs = torch.empty(nslabs * 1728, dtype=torch.float32, device="cuda")

    def bwd(fa):
        if fa:
            check(lib.vs_conv_k3_bwd_data_fused_apply(g.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), wpb.data_ptr(), y.data_ptr(),
                                                      mx.data_ptr(), mxs.data_ptr(), sums.data_ptr(), dx.data_ptr(), n, d, h, w, C, C, vdt, 1e-5, st), "fused apply")
        else:
            check(lib.vs_conv_gather_bwd_data(dx_ref.data_ptr(), wpb.data_ptr(), y.data_ptr(), mx.data_ptr(), mxs.data_ptr(), sums.data_ptr(),
                                              n, d, h, w, C, C, VS_CONV_K3, vdt, 1e-5, st), "bwd_data")

    def bwd_wgrad(fa):
        check(lib.vs_conv_k3_bwd_data_wgrad(g.data_ptr() if fa else dx_ref.data_ptr(), ax.data_ptr() if fa else None, axs.data_ptr() if fa else None,
                                            asums.data_ptr() if fa else None, wpb.data_ptr(), y.data_ptr(), mx.data_ptr(), mxs.data_ptr(), sums.data_ptr(),
                                            slabs.data_ptr(), n, d, h, w, C, C, vdt, 1e-5, st), "bwd_data + wgrad")

    def group(descs):
        arr = (ops.WgradDesc * len(descs))(*descs)
        nbytes = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(arr), len(descs), vdt)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        return lambda: check(lib.vs_conv_wgrad_multi(ctypes.addressof(arr), len(descs), ws.data_ptr(), nbytes, vdt, 1e-5, st), "conv_wgrad_multi")

    dws = [torch.empty(C, C, 27, dtype=torch.float32, device="cuda") for _ in range(3)]
    regular = [ops.WgradDesc(dx_ref.data_ptr(), None, mx.data_ptr(), mxs.data_ptr(), dws[i].data_ptr(), None, None, 0, 0, 0, n, d, h, w, C, C, C, C, VS_CONV_K3, 0)
               for i in range(3)]
    slab_d = [ops.WgradDesc(slabs.data_ptr(), None, None, None, dws[i].data_ptr(), None, None, 0, 0, 0, nslabs, 0, 0, 0, C, C, C, C, ops.VS_WGRAD_SLABS, 0)
              for i in range(3)]

    def timeit(fn, label):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        print("  %-100s %7.1f us" % (label, us))
        return us

    print("one 8 -> 8 3x3x3 layer at %d x %d^3, %s, %d slabs:" % (n, s, str(dt).replace("torch.", ""), nslabs))
    r = {}
    for fa in (False, True):
        tag = "un-applied gradient in (fused apply)" if fa else "applied gradient in"
        r[(fa, 0)] = timeit(lambda: bwd(fa), "backward-data, %s" % tag)
        r[(fa, 1)] = timeit(lambda: bwd_wgrad(fa), "backward-data + fused weight gradient, %s" % tag)
    bwd_wgrad(True)
    r["g1"] = timeit(group(regular[:1]), "grouped weight-gradient launch + reduction, this layer alone")
    r["g3"] = timeit(group(regular), "grouped weight-gradient launch + reduction, three such layers")
    r["s3"] = timeit(group(slab_d), "reduction of three layers' slabs (VS_WGRAD_SLABS descriptors only)")
    print("  per layer: the fused launch adds %.1f us (applied in) / %.1f us (fused apply); the grouped launch loses %.1f us, the slab reduction costs %.1f us"
          % (r[(False, 1)] - r[(False, 0)], r[(True, 1)] - r[(True, 0)], r["g3"] / 3, r["s3"] / 3))


if __name__ == "__main__":
    n, s = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 96)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dt = torch.float16 if (len(sys.argv) > 4 and sys.argv[4] == "fp16") else torch.bfloat16
    run(n, s, iters, dt)
    if len(sys.argv) <= 2:
        run(1, 128, iters, dt)
        run(1, 160, iters, torch.float16)
