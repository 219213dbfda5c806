"""GPU-box aid: cost of a device-wide barrier (flag per workgroup, everyone polls) for n resident workgroups — the price of fusing two
dependent launches into one.  usage: python tools/barrier_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import _lib  # noqa: F401  (loads torch's HIP runtime first)
from tools.probe import lib, check
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for mode, n in [(m, n) for m in (0, 1) for n in (8, 32, 72, 144, 256, 288, 432, 512)]:
    flags = torch.zeros(n, dtype=torch.int32, device="cuda")
    ticks = torch.zeros(n, dtype=torch.int64, device="cuda")
    check(lib.vs_debug_grid_barrier_probe(flags.data_ptr(), ticks.data_ptr(), n, iters, mode, None), "probe")
    torch.cuda.synchronize()
    t = ticks.cpu()
    if int(t.min()) < 0:
        print("%4d workgroups: not all resident (bounded spin gave up)" % n)
        continue
    print(("flags  " if mode == 0 else "counter") + " %4d workgroups: %.2f us per barrier (median workgroup; max %.2f)" % (n, float(t.median()) * 0.01 / iters, float(t.max()) * 0.01 / iters))
