"""GPU-box aid: the cost of a dependent launch — N one-wave kernels that exit at once (vs_spin(0)), as a replayed HIP graph and as
eager stream launches.  usage: python tools/launch_floor.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd._lib import lib, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        check(lib.vs_spin(0, st), "spin")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            check(lib.vs_spin(0, st), "spin")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("graph replay: %.2f us per empty kernel node (%d nodes)" % (e0.elapsed_time(e1) * 1e3 / (10 * n), n))
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        check(lib.vs_spin(0, st), "spin")
    e1.record()
    torch.cuda.synchronize()
    print("eager: %.2f us per launch on the GPU timeline, %.2f us of host time per launch" % (e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n))
