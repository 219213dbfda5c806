"""GPU-box aid: samples per second of the device data pipeline (vae_segmentation_amd/data_gpu.py) on a CT-sized synthetic case, next to
the numpy / scipy oracle (what the reference's CPU workers execute per sample: skimage resize + batchgenerators' map_coordinates).
usage: python tools/bench_data.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import data_cpu as O
from vae_segmentation_amd import data_gpu as D

rng = np.random.RandomState(0)
shape, patch = (200, 320, 320), (128, 128, 128)                    # an abdominal CT crop; main_source.py:117 patch_size 128^3
merge = np.zeros(shape + (2,), np.float32)
merge[..., 0] = rng.randn(*shape) * 300
merge[60:150, 100:230, 90:240, 1] = 1
p = O.draw_spatial_params(np.random.RandomState(1), patch, patch, [59] * 3)
t = D.MySpatialTransform(patch, [59] * 3, random_crop=True, scale=(0.85, 1.15), do_elastic_deform=False, angle_x=(-0.2, 0.2), angle_y=(-0.2, 0.2),
                         angle_z=(-0.2, 0.2), border_mode_data="constant", border_cval_data=-1024, data_key="venous", label_key="venous_pancreas",
                         p_el_per_sample=0)
mg = torch.from_numpy(merge).cuda()
par = (p["angles"], p["scale"], p["centre"], True)
for _ in range(3):
    D.train_sample(mg, patch, None, t, par)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    img, lab = D.train_sample(mg, patch, None, t, par)
torch.cuda.synchronize(); gpu_s = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
ref_i, ref_l = O.train_sample(merge, patch, p)
cpu_s = time.perf_counter() - t0
err = float(np.abs(img[0, 0].cpu().numpy() - ref_i).max())
rec = {"case": "merge %s -> CropResize(128^3) -> MySpatialTransform(rot/scale/crop, order 3) -> Clip -> CenterIntensities" % (shape,),
       "gpu_ms_per_sample": round(gpu_s * 1e3, 3), "gpu_samples_per_s": round(1.0 / gpu_s, 1),
       "cpu_oracle_s_per_sample_one_process": round(cpu_s, 2), "max_abs_err_vs_oracle": err,
       "label_mismatch_fraction": float((lab[0, 0].cpu().numpy() != ref_l).mean())}
print(json.dumps(rec))
if len(sys.argv) > 1:
    json.dump(rec, open(sys.argv[1], "w"), indent=1)
