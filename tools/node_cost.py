"""GPU-box aid: per-node cost of REAL small kernels replayed back to back on hot data (same arguments every node) — separates what a
kernel costs by itself from what cold data / a cold instruction cache add inside the step.  usage: python tools/node_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import lib, check, VS_BF16, VS_CONV_K3, VS_PACK_ROWS_D0, VS_PACK_ROWS_D1_FLIP

def time_graph(fn, n=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)

dt = torch.bfloat16
for side, c in ((6, 128), (12, 64), (24, 32)):
    n = 2
    x = torch.randn(n, side, side, side, c, device="cuda").to(dt)
    g = torch.randn_like(x)
    xs = ops.instnorm_stats(x)
    sums = torch.zeros(ops.STAT_SLOTS, n, c, 2, dtype=torch.float64, device="cuda")
    gx = torch.empty_like(x)
    vox = side ** 3
    st = lambda: torch.cuda.current_stream().cuda_stream
    t_apply = time_graph(lambda: check(lib.vs_instnorm_relu_bwd_apply(g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), gx.data_ptr(), n, vox, c, VS_BF16, 1e-5, st()), "a"))
    w = torch.randn(c, c, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, VS_PACK_ROWS_D0, c, dt)
    y = torch.empty_like(x)
    ys = torch.zeros(ops.STAT_SLOTS, n, c, 2, dtype=torch.float64, device="cuda")
    t_conv = time_graph(lambda: check(lib.vs_conv_gather_fwd(x.data_ptr(), xs.data_ptr(), wp.data_ptr(), None, y.data_ptr(), ys.data_ptr(), n, side, side, side, c, c, VS_CONV_K3, VS_BF16, 1e-5, st()), "c"))
    wpb = ops.pack_weight(w, VS_PACK_ROWS_D1_FLIP, c, dt)
    t_bwd = time_graph(lambda: check(lib.vs_conv_gather_bwd_data(g.data_ptr(), wpb.data_ptr(), gx.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), n, side, side, side, c, c, VS_CONV_K3, VS_BF16, 1e-5, st()), "b"))
    print("%2d^3 x %3d ch: apply %.2f us, conv fwd %.2f us, conv bwd-data+sums %.2f us per node (hot data, same kernel back to back)" % (side, c, t_apply, t_conv, t_bwd))
