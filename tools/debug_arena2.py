import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_model as M
from oracle import ref_cpu as O
from vae_segmentation_amd import ops, optim, train as T
seg = M.Segmentation(1, 2, norm_type=1); vae = M.VAE(2, 2, norm_type=1, dim=128, spatial=96)
j = M.Joint([seg, vae]); O.deterministic_fill_(j, 0); j = j.cuda()
for p in j.Vae.parameters(): p.requires_grad = False
M.set_kernel_dtype(j, torch.bfloat16)
img, lab = O.synthetic_image(2, 96, 2).cuda(), O.synthetic_label(2, 96, 3).cuda()
opt = optim.SGD(j.Seg.parameters(), lr=1e-2, momentum=0.9)
orig = ops._new_stats
log = []
def spy(n, c, device, width=2):
    a = ops._ARENA
    before = a.get("fallbacks", 0)
    out = orig(n, c, device, width)
    if a.get("fallbacks", 0) != before:
        log.append((n, c, width, a["off"], a["buf"].numel() if a["buf"] is not None else -1, str(a["buf"].device) if a["buf"] is not None else None, str(device)))
    return out
ops._new_stats = spy
gs = T.GraphedStep(lambda: T.joint_train_losses(j, img, lab), list(j.Seg.parameters()), opt, warmup=2)
print("fallbacks total", ops._ARENA.get("fallbacks"), "logged", len(log))
for l in log[:5]: print(l)
for l in log[-5:]: print(l)
