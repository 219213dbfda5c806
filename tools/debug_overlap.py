import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_model as M
from oracle import ref_cpu as O
from vae_segmentation_amd import ops, optim, train as T
def grads(overlap, graph=False):
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), 0).cuda()
    img, lab = O.synthetic_image(2, 32, 2).cuda(), O.synthetic_label(2, 32, 3).cuda()
    ops.set_overlap(overlap)
    opt = optim.SGD(seg.parameters(), lr=1e-2, momentum=0.9)
    if graph:
        gs = T.GraphedStep(lambda: T.seg_train_losses(seg, img, lab), list(seg.parameters()), opt, warmup=1, overlap=overlap)
        gs.graph.replay(); torch.cuda.synchronize()
        return [p.grad.clone() for p in seg.parameters()], gs.loss.item()
    l, _ = T.seg_train_losses(seg, img, lab); l.backward(); ops.join_side(); torch.cuda.synchronize()
    return [p.grad.clone() for p in seg.parameters()], l.item()
g0, l0 = grads(False)
for name, (ov, gr) in {"eager+overlap": (True, False), "graph": (False, True), "graph+overlap": (True, True)}.items():
    g1, l1 = grads(ov, gr)
    worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(g0, g1))
    print("%-14s loss %.7f (ref %.7f) worst grad rel diff %.3e" % (name, l1, l0, worst))
