"""GPU-box aid: the cost of a dependent launch — N one-wave kernels that exit at once (vs_spin(0)), as a replayed HIP graph and as
eager stream launches.  usage: python tools/launch_floor.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_segmentation_amd import _lib  # noqa: F401  (loads torch's HIP runtime first)
from tools.probe import lib, check
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

    buf = torch.zeros(65536 * 256 + 8, device="cuda")
    for nwg in (1, 64, 1024):
        for mode, name in ((0, "plain store"), (1, "non-temporal store"), (2, "atomic store, agent scope"), (3, "atomic store, system scope"), (4, "load only"), (5, "p[i] += 1 (own element)"), (6, "p[i] = p[other] + 1 (cross-XCD)")):
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=s):
                st2 = torch.cuda.current_stream().cuda_stream
                for _ in range(n):
                    check(lib.vs_debug_store_probe(buf.data_ptr(), nwg, mode, st2), "probe")
            for _ in range(3):
                g2.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                g2.replay()
            e1.record()
            torch.cuda.synchronize()
            print("graph node, %4d workgroups, %-28s %.2f us" % (nwg, name + ":", e0.elapsed_time(e1) * 1e3 / (10 * n)))

    # alternating two different kernels (instruction-cache effect?) with a true dependency
    for nwg in (1, 64, 1024):
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, stream=s):
            st3 = torch.cuda.current_stream().cuda_stream
            for i in range(n):
                if i % 2:
                    check(lib.vs_debug_store_probe(buf.data_ptr(), nwg, 6, st3), "probe")
                else:
                    check(lib.vs_zero_async_probe(buf.data_ptr(), nwg * 256 * 4, st3) if hasattr(lib, "vs_zero_async_probe") else lib.vs_spin(0, st3), "x")
        for _ in range(3):
            g3.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            g3.replay()
        e1.record()
        torch.cuda.synchronize()
        print("graph node, %4d workgroups, alternating spin(0) / cross-XCD rmw: %.2f us per node" % (nwg, e0.elapsed_time(e1) * 1e3 / (10 * n)))
