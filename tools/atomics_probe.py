"""GPU-box aid: what do the per-(n,c) statistics atomics cost?  Every conv layer shape of the 96^3 step, forward, replayed back to back in a HIP
graph, once accumulating (sum, sumsq) (y_stats given: one fp64 atomic per workgroup per channel per statistic) and once without.
usage: python tools/atomics_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import lib, check, VS_BF16, VS_CONV_K3, VS_PACK_ROWS_D0


def time_graph(fn, n=100):
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
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


dt = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
for (n, cin, cout, side) in [(2, 8, 8, 96), (2, 16, 8, 96), (2, 8, 16, 48), (2, 16, 16, 48), (2, 32, 16, 48), (2, 16, 32, 24), (2, 32, 32, 24),
                             (2, 64, 32, 24), (2, 32, 64, 12), (2, 64, 64, 12), (2, 128, 128, 6), (2, 256, 256, 3)]:
    x = torch.randn(n, side, side, side, cin, device="cuda").to(dt)
    xs = ops.instnorm_stats(x)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, VS_PACK_ROWS_D0, cin, dt)
    y = torch.empty(n, side, side, side, cout, device="cuda", dtype=dt)
    ys = torch.zeros(ops.STAT_SLOTS, n, cout, 2, dtype=torch.float64, device="cuda")
    res = []
    for stats in (ys.data_ptr(), None):
        res.append(time_graph(lambda: check(lib.vs_conv_gather_fwd(x.data_ptr(), xs.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stats, n, side, side, side,
                                                                   cin, cout, VS_CONV_K3, VS_BF16, 1e-5, st()), "c")))
    print("%3d -> %3d @ %2d^3: with statistics %6.2f us, without %6.2f us  (atomics + reduction: %5.2f us)" % (cin, cout, side, res[0], res[1], res[0] - res[1]), flush=True)
