"""experiment: does running the two samples of the batch as two parallel branches of one HIP graph pay?  forward only (no_grad), joint net"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_model as M
from oracle import ref_cpu as O
side = int(sys.argv[1]) if len(sys.argv) > 1 else 96
seg = M.Segmentation(1, 2, norm_type=1); vae = M.VAE(2, 2, norm_type=1, dim=128, spatial=side)
joint = M.Joint(models=[seg, vae]); O.deterministic_fill_(joint, seed=0); joint = joint.cuda()
for p in joint.parameters(): p.requires_grad = False
M.set_kernel_dtype(joint, torch.bfloat16)
img = O.synthetic_image(2, side, 2).cuda()
parts = [img[0:1].contiguous(), img[1:2].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def batched():
    with torch.no_grad():
        return joint({"x": img}, "x", "p", "r")["r"]

def split(n_streams):
    outs = []
    main = torch.cuda.current_stream()
    with torch.no_grad():
        for i in range(2):
            s = streams[i % n_streams]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(joint({"x": parts[i]}, "x", "p", "r")["r"])
        for s in streams[:n_streams]:
            main.wait_stream(s)
    return outs

def bench(fn, name):
    side_s = torch.cuda.Stream(); side_s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side_s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side_s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("%-28s %.3f ms per forward of 2 volumes" % (name, dt * 1e3), flush=True)
    return out

a = bench(batched, "batched N=2, one stream")
b = bench(lambda: split(1), "per-sample, one stream")
c = bench(lambda: split(2), "per-sample, two streams")
print("max diff split vs batched", float((torch.cat(c) - a).abs().max()))
