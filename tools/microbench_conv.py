"""GPU-box aid: time one conv shape in isolation (for rocprofv3 --pmc runs).
usage: python tools/microbench_conv.py N C M S [iters] [dtype] [lazy]"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import VS_CONV_K3, VS_PACK_ROWS_D0

n, c, m, s = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dtype = torch.float32 if (len(sys.argv) > 6 and sys.argv[6] == "fp32") else torch.bfloat16
lazy = len(sys.argv) > 7 and sys.argv[7] == "lazy"
x = torch.randn(n, s, s, s, c, device="cuda").to(dtype)
w = torch.randn(m, c, 3, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, VS_PACK_ROWS_D0, c, dtype)
xs = ops.instnorm_stats(x) if lazy else None
for _ in range(3):
    ops.conv_gather(x, xs, wp, None, m, VS_CONV_K3, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.conv_gather(x, xs, wp, None, m, VS_CONV_K3, True)
e1.record()
torch.cuda.synchronize()
print("conv k3 n=%d c=%d m=%d s=%d %s lazy=%s: %.1f us per call (incl. stats memset)" % (n, c, m, s, dtype, lazy, 1e3 * e0.elapsed_time(e1) / iters))
