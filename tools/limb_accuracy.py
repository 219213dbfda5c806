"""GPU-box aid: end-to-end gradient accuracy of the fp32 parity mode with the limb kernels on / off (run twice: VS_F32_LIMBS=1 / 0).
Seg 32^3 and joint 64^3 against the reference goldens' fp64 yardstick: per-tensor error, worst five, and the per-op error of one conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import joint_model as M
from oracle import ref_cpu as O
from tests import golden_util as G
from vae_segmentation_amd import ops, train as T

print("VS_F32_LIMBS =", os.environ.get("VS_F32_LIMBS", "(default 1)"), " library says", ops._F32_LIMBS)
g = G.load("seg32")
seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
img, lab = O.synthetic_image(2, 32, 2), O.synthetic_label(2, 32, 3)
loss, aux = T.seg_train_losses(seg, img.cuda(), lab.cuda(), eps=1e-6)
loss.backward()
rep = G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=1.0)
rep.sort(key=lambda r: -r[1])
print("seg32: loss %.9f (golden fp64 %.9f)" % (loss.item(), float(g["dice_loss_eps1e6@f64"])))
print("seg32 gradient error vs fp64 (mine / reference-fp32): worst five:", [(n, "%.2e" % m, "%.2e" % t) for n, m, t, _ in rep[:5]])
print("seg32 median mine %.2e  median reference %.2e" % (np.median([r[1] for r in rep]), np.median([r[2] for r in rep])))
# per-op error: one 16 -> 16 conv at 24^3 against fp64
torch.manual_seed(0)
x = torch.randn(2, 16, 24, 24, 24)
w = torch.randn(16, 16, 3, 3, 3) * 0.05
ref = F.conv3d(x.double(), w.double(), padding=1)
x_cl = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
ops.stats_arena_begin(x_cl.device)
y, _ = ops.ConvK3.apply(x_cl, None, w.cuda(), None)
yy = y.permute(0, 4, 1, 2, 3).double().cpu()
e = (yy - ref)
print("conv 16->16 @24^3: rel l2 %.3e, max %.3e (of max |y|), mean signed error / rms %.3e" % (float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()),
                                                                                               float(e.mean() / ref.pow(2).mean().sqrt())))
y32 = F.conv3d(x, w, padding=1).double()
e2 = y32 - ref
print("torch CPU fp32 conv:  rel l2 %.3e, max %.3e" % (float(e2.norm() / ref.norm()), float(e2.abs().max() / ref.abs().max())))

# how much of the end-to-end distance is the network's rounding amplification?  the SAME kernels, inputs perturbed by one fp32 ulp at random:
# the spread of the per-tensor errors between two such runs is the size of a "draw"
for trial in range(3):
    seg2 = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    gen = torch.Generator(device="cuda").manual_seed(100 + trial)
    with torch.no_grad():
        for p_ in seg2.parameters():
            p_.mul_(1.0 + (torch.rand(p_.shape, device="cuda", generator=gen) - 0.5) * 2.4e-7)      # +- 1 ulp
    l2, _ = T.seg_train_losses(seg2, img.cuda(), lab.cuda(), eps=1e-6)
    l2.backward()
    rep2 = G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg2.named_parameters()], floor=1.0)
    rep2.sort(key=lambda r: -r[1])
    print("weights perturbed by +-1 ulp, trial %d: median error %.2e, worst %s %.2e" % (trial, np.median([r[1] for r in rep2]), rep2[0][0], rep2[0][1]))
