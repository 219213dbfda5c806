"""GPU-box aid: Embed at 128^3 (three chained networks: tests/test_gpu_model.py::test_embed128_vs_reference_golden) — per-tensor gradient error against
the reference's fp64 run with the limb kernels on / off (run twice: VS_F32_LIMBS=1 / 0), unperturbed and with the weights perturbed by +-1 fp32 ulp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import joint_model as M
from oracle import ref_cpu as O
from tests import golden_util as G
from vae_segmentation_amd.evaluation import avg_dsc

g = G.load("embed128")
print("VS_F32_LIMBS =", os.environ.get("VS_F32_LIMBS", "(default 1)"))
img, gt = O.synthetic_image(1, 128, seed=2).cuda(), O.one_hot(O.synthetic_label(1, 128, seed=3)).cuda()
for trial in range(3):
    emb = M.Embed(models=[M.Encoder(1, 128, norm_type=1), M.VAE(2, 2, norm_type=1, dim=128), M.Fusion(1, 2, 2, norm_type=1)])
    O.deterministic_fill_(emb, seed=8)
    emb = emb.cuda()
    if trial:
        gen = torch.Generator(device="cuda").manual_seed(200 + trial)
        with torch.no_grad():
            for p_ in emb.parameters():
                p_.mul_(1.0 + (torch.rand(p_.shape, device="cuda", generator=gen) - 0.5) * 2.4e-7)
    batch = emb({"img": img, "venous_pancreas_only": gt, "gt": gt}, "img", "pred", noise=torch.from_numpy(g["z"]).cuda())
    dsc = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-4)
    lat = torch.mean((batch["latent_code"] - batch["latent_code_gt"].detach()) ** 2)
    (dsc + lat).backward()
    rows = []
    for pre, mod in (("enc", emb.Encoder), ("vae", emb.Vae), ("fus", emb.Fusion)):
        rep = G.check_grads_f64(g, pre, [(n, p.grad) for n, p in mod.named_parameters()], floor=10.0, hard_factor=1e9)
        rows += [(pre + "." + n, m, t) for n, m, t, _ in rep]
    rows.sort(key=lambda r: -r[1] / max(r[2], 1e-30))
    med = np.median([r[1] for r in rows]); medr = np.median([r[2] for r in rows])
    print("%s: %d tensors, median error %.2e (reference fp32 %.2e); worst ratios to the reference's own error: %s" % (
        "unperturbed" if not trial else "weights +-1 ulp, trial %d" % trial, len(rows), med, medr,
        [(n, "%.1e" % m, "%.1e" % t, "x%.0f" % (m / t)) for n, m, t in rows[:4]]))
    del emb
