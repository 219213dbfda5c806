"""A LEARNABLE synthetic segmentation task for the convergence checks (tests/test_gpu_convergence.py, tools/convergence.py).

The benchmark's inputs (oracle/ref_cpu.py: synthetic_image = clipped noise, independent of the label) time a step but cannot be learnt.
Here the image carries the label: every volume is one ellipsoid with its own centre and radii, the image is
clip(contrast * (2 * label - 1) smoothed over the rim + sigma * N(0,1), -1, 1) — the value range `Clip` / `CenterIntensities` leave behind
(/root/reference/main_source.py:211-212).  Seeded torch.Generator on the CPU: the same volumes on every box and for every precision mode.
"""
import torch


def volumes(count, side, seed=0, contrast=0.45, sigma=0.6):
    """-> (img [count,1,S,S,S] fp32 in [-1,1], label [count,1,S,S,S] fp32 in {0,1})"""
    g = torch.Generator().manual_seed(1000 + seed)
    ax = (torch.arange(side, dtype=torch.float32) + 0.5) / side - 0.5
    z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
    imgs, labs = [], []
    for _ in range(count):
        c = (torch.rand(3, generator=g) - 0.5) * 0.3                       # centre within +-0.15 of the middle
        r = 0.18 + 0.2 * torch.rand(3, generator=g)                        # radii 0.18 .. 0.38 of the side
        d = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2
        lab = (d < 1.0).float()
        soft = torch.clamp((1.0 - d) * 4.0, -1.0, 1.0)                     # +-1 away from the rim, a ramp across it
        img = torch.clamp(contrast * soft + sigma * torch.randn(side, side, side, generator=g), -1.0, 1.0)
        imgs.append(img)
        labs.append(lab)
    return torch.stack(imgs)[:, None].contiguous(), torch.stack(labs)[:, None].contiguous()
