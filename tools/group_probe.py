"""GPU-box aid: price of the group protocol an in-epilogue InstanceNorm-backward apply would need (csrc/misc.hip, grid_barrier_probe_kernel modes 2 / 3):
partial sums by fp64 atomics (done today), then group counter + poll + atomic loads of the totals.  usage: python tools/group_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import _lib  # noqa: F401  (loads torch's HIP runtime first)
from tools.probe import lib, check
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n, gsz in [(32, 1), (64, 4), (72, 9), (144, 9), (288, 72), (288, 144), (256, 256)]:
    res = {}
    for mode in (3, 2):
        flags = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
        ticks = torch.zeros(n + (n // gsz + 1) * 64, dtype=torch.int64, device="cuda")        # the sums region is zero doubles
        check(lib.vs_debug_grid_barrier_probe(flags.data_ptr(), ticks.data_ptr(), n, iters, mode | (gsz << 8), None), "probe")
        torch.cuda.synchronize()
        t = ticks[:n].cpu()
        res[mode] = None if int(t.min()) < 0 else float(t.median()) * 0.01 / iters
    if res[2] is None or res[3] is None:
        print("%4d workgroups in groups of %3d: bounded spin gave up" % (n, gsz))
    else:
        print("%4d workgroups in groups of %3d: sums alone %.2f us, with counter + poll + read-back %.2f us per round -> +%.2f us" % (n, gsz, res[3], res[2], res[2] - res[3]))
