"""GPU-box aid: time the composed Up head (ops.UpConvK3: k4t_kernel forward, k4g_kernel backward-data) and the two-launch pair it replaces, in isolation.
usage: python tools/microbench_up.py N C Co S [iters]      (S = coarse side)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import VS_CONV_K3, VS_CONV_K2S2, VS_PACK_ROWS_D0, VS_PACK_ROWS_D1_FLIP, VS_PACK_SCATTER_D1, lib, check

n, c, co, s = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dt = torch.bfloat16
x = torch.randn(n, s, s, s, c, device="cuda").to(dt)
xs = ops.instnorm_stats(x)
wt = torch.randn(c, c, 2, 2, 2, device="cuda") * 0.2
bt = torch.randn(c, device="cuda") * 0.1
w3 = torch.randn(co, c, 3, 3, 3, device="cuda") * 0.05
gy = torch.randn(n, 2 * s, 2 * s, 2 * s, co, device="cuda").to(dt)
plan = ops.up_plan(wt, bt, w3, dt)
y = torch.empty(n, 2 * s, 2 * s, 2 * s, co, device="cuda", dtype=dt)
gx = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-46s %.1f us" % (label, 1e3 * e0.elapsed_time(e1) / iters))


ys = torch.zeros(4, n, co, 2, dtype=torch.float64, device="cuda")
sums = torch.zeros(4, n, c, 2, dtype=torch.float64, device="cuda")
def fwd():
    check(lib.vs_up_conv_fwd(x.data_ptr(), xs.data_ptr(), plan["img_f"].data_ptr(), plan["taps_f"].data_ptr(), plan["btab"].data_ptr(), y.data_ptr(),
                             ys.data_ptr(), n, s, s, s, c, co, 1, 1e-5, st), "fwd")
def bwd():
    check(lib.vs_up_conv_bwd_data(gy.data_ptr(), plan["img_b"].data_ptr(), plan["taps_b"].data_ptr(), gx.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(),
                                  n, s, s, s, co, c, 1, 1e-5, st), "bwd")
timeit(fwd, "composed fwd  x(%d,%d^3,%d)->%d" % (n, s, c, co))
timeit(bwd, "composed bwd-data")
# the pair it replaces
wps = ops.pack_weight(wt, VS_PACK_SCATTER_D1, c, dt)
wp3 = ops.pack_weight(w3, VS_PACK_ROWS_D0, c, dt)
u = [None]
def pair_fwd():
    u[0] = ops.conv_scatter(x, xs, wps, bt, c)
    ops.conv_gather(u[0], None, wp3, None, co, VS_CONV_K3, True)
ops.stats_arena_begin(x.device)
timeit(pair_fwd, "two-launch fwd (scatter + 3x3x3)")
