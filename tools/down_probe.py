"""GPU-box aid ("measure the composed Down head"): Conv3d(C, C, 2, stride 2) -> Conv3d(C, Co, 3, padding 1) (joint_model.py:126-136,
nothing in between) as ONE 6x6x6 / stride-2 operator (tools/probe/probe.hip: down_composed_probe_kernel) against the two launches the library runs for it.
Forward only, bf16, C = 16 -> Co = 32 (the 48^3 -> 24^3 head of configs[1]).   usage: python tools/down_probe.py [N S iters]   (S = fine side)"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import VS_CONV_K3, VS_CONV_K2S2, VS_PACK_ROWS_D0
from tools import probe

n, s = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 48)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
C, CO = 16, 32
dt = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, s, s, C, generator=g).to(dt).cuda()
w2 = (torch.randn(C, C, 2, 2, 2, generator=g) * 0.2).cuda()
w3 = (torch.randn(CO, C, 3, 3, 3, generator=g) * 0.08).cuda()
# Weff[co][ci][pz][py][px] = sum over (d, t) with p = 2 d + t of W3[co][cm][d] W2[cm][ci][t]: per axis p in 0..5 has exactly one (d, t) = (p // 2, p % 2)
weff = torch.zeros(CO, C, 6, 6, 6, device="cuda")
for pz in range(6):
    for py in range(6):
        for px in range(6):
            d, t = (pz // 2, py // 2, px // 2), (pz % 2, py % 2, px % 2)
            weff[:, :, pz, py, px] = w3[:, :, d[0], d[1], d[2]] @ w2[:, :, t[0], t[1], t[2]]
weff = weff.to(dt).float()                                   # ONE rounding to the storage type, as the composed Up head does
# fragment image [row block][k-group][lane][8]: row = 16 rb + (lane & 15); k-group kg = taps (2 kg, 2 kg + 1); lane group g = lane >> 4: tap 2 kg + (g >> 1), channels 8 (g & 1) ..
wflat = weff.reshape(CO, C, 216)
img = torch.empty(2, 108, 64, 8, device="cuda")
lane = torch.arange(64, device="cuda")
for rb in range(2):
    rows = rb * 16 + (lane & 15)
    gg = lane >> 4
    for kg in range(108):
        tap = 2 * kg + (gg >> 1)
        for j in range(8):
            img[rb, kg, :, j] = wflat[rows, (gg & 1) * 8 + j, tap]
img = img.to(dt).contiguous()
y = torch.empty(n, s // 2, s // 2, s // 2, CO, device="cuda", dtype=dt)
st = torch.cuda.current_stream().cuda_stream


def composed():
    probe.check(probe.lib.vs_debug_down_composed_probe(x.data_ptr(), img.data_ptr(), y.data_ptr(), n, s, s, s, st), "down_composed_probe")


composed()
torch.cuda.synchronize()
ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), weff, stride=2, padding=2).permute(0, 2, 3, 4, 1)
err = float((y.float() - ref).norm() / ref.norm())
print("composed 6x6x6 / stride-2 operator vs F.conv3d on the same rounded Weff: relative L2 %.2e" % err)
assert err < 1e-2


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
    us = 1e3 * e0.elapsed_time(e1) / iters
    print("%-78s %.1f us" % (label, us))
    return us


t_c = timeit(composed, "composed Down head, ONE launch (x (%d, %d^3, %d) -> (%d^3, %d)), no normalise-on-load" % (n, s, C, s // 2, CO))
wp2 = ops.pack_weight(w2, VS_PACK_ROWS_D0, C, dt)
wp3 = ops.pack_weight(w3, VS_PACK_ROWS_D0, C, dt)
b2 = torch.zeros(C, device="cuda")
ops.stats_arena_begin(x.device)
xs = ops.instnorm_stats(x)


def pair():
    u, _ = ops.conv_gather(x, xs, wp2, b2, C, VS_CONV_K2S2, False)          # the library's stride-2 conv (lazy input: normalise + ReLU on load, as in the step)
    ops.conv_gather(u, None, wp3, None, CO, VS_CONV_K3, True)


t_p = timeit(pair, "the library's two launches (g1_kernel<K2S2> + k3b_kernel), lazy input, statistics out")
flops_c, flops_p = 2.0 * n * (s // 2) ** 3 * 216 * C * CO, 2.0 * n * (s // 2) ** 3 * (8 * C * C + 27 * C * CO)
print("MACs per output voxel: composed %d, pair %d (%.1f x); composed %.1f TFLOP/s" % (216 * C * CO, 8 * C * C + 27 * C * CO, flops_c / flops_p, flops_c / t_c / 1e6))
print("verdict: composed is %.2f x the pair's time" % (t_c / t_p))
