"""GPU-box aid: latency of one test-time-training case (main_target.py:809-953): weight reset, k iterations
(student fwd+bwd, teacher fwd, SGD), validation forwards of both networks + hard Dice.  usage: bench_finetune.py [size] [k] [dtype] [graph]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_model as M
from oracle import ref_cpu as O          # deterministic weights / synthetic inputs only
from vae_segmentation_amd import train as T
from vae_segmentation_amd.modules import set_kernel_dtype

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
graph = not (len(sys.argv) > 4 and sys.argv[4] == "eager")


def joint(seed):
    j = M.Joint(models=[M.Segmentation(1, 2, norm_type=1), M.VAE(2, 2, norm_type=1, dim=128, spatial=size)])
    O.deterministic_fill_(j, seed=0)
    if seed:
        O.deterministic_fill_(j.Seg, seed=seed)
    j = j.cuda()
    for p in j.Vae.parameters():
        p.requires_grad = False
    set_kernel_dtype(j, dtype)
    return j


model, model_ft, teacher = joint(0), joint(0), joint(1)
for p in teacher.parameters():
    p.requires_grad = False
runner = T.TestTimeFinetune(model, model_ft, teacher, size, steps=k, lr=1e-2, lambda_vae=1.0, domain_loss_type=8, graph=graph)
img, lab = O.synthetic_image(1, size, 2).cuda(), O.synthetic_label(1, size, 3).cuda()
for _ in range(3):
    runner.run(img, lab)
torch.cuda.synchronize()
t0 = time.time()
n = 20
for _ in range(n):
    log, s0, s1, _ = runner.run(img, lab)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print("test-time finetune %d^3 k=%d %s %s: %.2f ms per case (%.1f cases/s); last losses: %s; dice noft %.4f ft %.4f" % (
    size, k, "bf16" if dtype == torch.bfloat16 else "fp32", "graph" if graph else "eager", dt * 1e3, 1 / dt,
    {kk: round(v.item(), 4) for kk, v in log[-1].items()}, s0.item(), s1.item()))
