"""GPU-box debugging aid: stage-by-stage comparison of the native VAE against the CPU oracle."""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import joint_model as M
from oracle import ref_cpu as O
from vae_segmentation_amd import ops
from vae_segmentation_amd.modules import Act

side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.set_num_threads(32)
ov = O.deterministic_fill_(O.VAE(2, 2, norm_type=1, dim=128, spatial=side), seed=0)
mv = O.deterministic_fill_(M.VAE(2, 2, norm_type=1, dim=128, spatial=side), seed=0).cuda()
gt = O.one_hot(O.synthetic_label(bs, side, 3))


def cmp(name, a_cl, ref):
    a = ops.UnpackPlanar.apply(ops.Materialize.apply(a_cl.raw, a_cl.stats, None, None), ref.shape[1]).cpu()
    err = (a - ref).abs().max().item() / ref.abs().max().item()
    print("%-10s shape %-22s max-rel-err %.3e" % (name, tuple(ref.shape), err))


with torch.no_grad():
    a = Act(ops.PackPlanar.apply(gt.cuda(), torch.float32), None)
    r = gt
    a, r = mv.in_block(a), ov.in_block(r); cmp("in_block", a, r)
    for i in range(1, 6):
        a, r = getattr(mv, "down%d" % i)(a), getattr(ov, "down%d" % i)(r); cmp("down%d" % i, a, r)
    feat = ops.Materialize.apply(a.raw, a.stats, None, None)
    mean = ops.LinearCL.apply(feat, mv.fc_mean.weight, mv.fc_mean.bias, False)
    std = ops.LinearCL.apply(feat, mv.fc_std.weight, mv.fc_std.bias, True)
    rf = r.reshape(bs, -1)
    rmean, rstd = ov.fc_mean(rf), torch.relu(ov.fc_std(rf))
    print("mean err %.3e std err %.3e" % ((mean.cpu() - rmean).abs().max() / rmean.abs().max(), (std.cpu() - rstd).abs().max() / rstd.abs().max()))
    noise = torch.from_numpy(2 * O.hashed_uniform(bs * 128, 7100, 5) - 1).view(bs, 128)
    z = ops.Reparam.apply(mean, std, noise.cuda(), 0.35)
    rz = rmean + noise * rstd * 0.35
    print("z err %.3e" % ((z.cpu() - rz).abs().max() / rz.abs().max()))
    h = ops.LinearToCL.apply(z, mv.fc2.weight, mv.fc2.bias, 256, side // 32, torch.float32)
    rh = ov.fc2(rz).view(bs, 256, side // 32, side // 32, side // 32)
    a = Act(h, None); cmp("fc2", a, rh)
    r = rh
    for i in range(1, 6):
        a, r = getattr(mv, "up%d" % i)(a), getattr(ov, "up%d" % i)(r); cmp("up%d" % i, a, r)
    recon = ops.ConvK3Softmax.apply(a.raw, a.stats, mv.out_block.weight, mv.out_block.bias)
    rrec = ov.final(ov.out_block(r))
    e = (recon.cpu() - rrec).abs()
    print("recon max err %.3e mean err %.3e; frac>1e-3: %.4f" % (e.max(), e.mean(), (e > 1e-3).float().mean()))
