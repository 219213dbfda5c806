import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_model as M
from oracle import ref_cpu as O
from vae_segmentation_amd import ops, optim, train as T
seg = M.Segmentation(1, 2, norm_type=1); vae = M.VAE(2, 2, norm_type=1, dim=128, spatial=64)
j = M.Joint([seg, vae]); O.deterministic_fill_(j, 0); j = j.cuda()
for p in j.Vae.parameters(): p.requires_grad = False
M.set_kernel_dtype(j, torch.bfloat16)
img, lab = O.synthetic_image(2, 64, 2).cuda(), O.synthetic_label(2, 64, 3).cuda()
opt = optim.SGD(j.Seg.parameters(), lr=1e-2, momentum=0.9)
for it in range(4):
    ops._ARENA["fallbacks"] = 0
    for p in j.Seg.parameters(): p.grad = None
    l, _ = T.joint_train_losses(j, img, lab); l.backward(); opt.step()
    torch.cuda.synchronize()
    a = ops._ARENA
    print(it, "loss %.5f" % l.item(), "fallbacks", a["fallbacks"], "need", a["need"], "used", a["used"], "off", a["off"], "cap", a["buf"].numel())
