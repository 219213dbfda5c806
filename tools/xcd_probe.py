"""GPU-box aid (round 6): price of ONE per-sample layer boundary kept inside a launch — payload stores, 32 fp64 partial
statistics per workgroup, group counter, poll, read-back (tools/probe/probe.hip: xcd_group_probe_kernel) — against the same round as a dependent
launch of a captured chain.  Groups of 8 / 16 / 32 (/ 4 / 9) workgroups stand for the workgroups of one sample's layer at <= 12^3; `same XCD` groups
have equal blockIdx % 8.  usage: python tools/xcd_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import _lib  # noqa: F401  (loads torch's HIP runtime first)
from tools.probe import lib, check
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = "cuda"
NAMES = {0: "spread, sc1 stores, sc1 loads", 1: "same XCD, sc1 stores, sc1 loads", 3: "same XCD, plain stores, sc1 loads", 7: "same XCD, plain stores, sc0 loads",
         9: "same XCD, sc1 stores, sc1 loads, stats by fp64 atomics", 16: "spread, plain + release / acquire fences", 17: "same XCD, plain + release / acquire fences",
         2: "spread, plain stores, sc1 loads"}


def run(n, gsz, pb, mode):
    flags = torch.zeros(n * 32, dtype=torch.int32, device=dev)
    ticks = torch.zeros(n, dtype=torch.int64, device=dev)
    xcc = torch.zeros(n, dtype=torch.int32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    payload = torch.zeros(2 * n * pb, dtype=torch.uint8, device=dev)
    slots = torch.zeros(2 * n * 32, dtype=torch.float64, device=dev)
    check(lib.vs_debug_xcd_group_probe(flags.data_ptr(), ticks.data_ptr(), xcc.data_ptr(), err.data_ptr(), payload.data_ptr(), slots.data_ptr(), n, iters, gsz, pb, mode, None), "xcd probe")
    torch.cuda.synchronize()
    t = ticks.cpu()
    x = xcc.cpu()
    same = bool(((x.view(-1, 8) - x[:8].view(1, 8)) == 0).all())          # blocks b and b + 8 on one XCD?
    if int(t.min()) < 0:
        return None, int(err[0]), same
    return float(t.median()) * 0.01 / iters, int(err[0]), same


def chain(n, gsz, pb, k=200):
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    payload = torch.zeros(2 * n * pb, dtype=torch.uint8, device=dev)
    slots = torch.zeros(2 * n * 32, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in range(1, 4):
            check(lib.vs_debug_xcd_chain_probe(err.data_ptr(), payload.data_ptr(), slots.data_ptr(), n, gsz, pb, it, s.cuda_stream), "chain")
        s.synchronize()
        err.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for it in range(1, k + 1):
                check(lib.vs_debug_xcd_chain_probe(err.data_ptr(), payload.data_ptr(), slots.data_ptr(), n, gsz, pb, it, s.cuda_stream), "chain")
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (5 * k), int(err[0])


print("per-sample layer boundary inside one launch vs a dependent launch; %d rounds; us per round (median workgroup); stale = payload / partial words that were not this round's" % iters)
for n, gsz in [(64, 8), (128, 16), (256, 32), (256, 8), (32, 4), (72, 9)]:
    for pb in (4096, 16384):
        c, ce = chain(n, gsz, pb)
        print("\n%3d workgroups, groups of %2d, %2d KB payload per workgroup:   dependent launches in a graph %.2f us per launch (stale %d)" % (n, gsz, pb >> 10, c, ce))
        for mode in (1, 0, 3, 2, 7, 9, 17, 16):
            if (mode & 1) and (n % 8 or (n // 8) % gsz):
                continue
            r, e, same = run(n, gsz, pb, mode)
            print("   %-58s %s   stale %d%s" % (NAMES[mode], ("%.2f us" % r) if r is not None else "bounded spin gave up", e, "" if same else "   (placement NOT round-robin)"))
