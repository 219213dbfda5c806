"""RNG-free weight fill and synthetic volumes of the benchmark workloads (bench.py, tools/): pure functions of (seed, stream, index), so every rank,
run and box sees the same numbers and nothing depends on a generator's version.  Product code — bench.py's set-up must not lean on test infrastructure
 — holding the SAME counter hash as the oracle's fixtures use (tests/test_host.py::test_synthetic_matches_the_oracle_generators
compares them value for value), so the benchmark's `final_loss` stays comparable across rounds.  Shapes follow main_source.py:211-212
(image ~ clip(N(0, 1), -1, 1) after Clip / CenterIntensities) and :449-451 (integer label volume)."""
import math
import zlib

import numpy as np
import torch


def _mix64(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hashed_uniform(n, stream, seed=0):
    """n floats in [0, 1): 24 bits of the splitmix64 finaliser of a per-(seed, stream) key plus the golden-ratio multiple of the index"""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) % (1 << 64))
        h = _mix64(idx * np.uint64(0x9E3779B97F4A7C15) + key)
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def deterministic_fill_(module, seed=0, gain=1.0):
    """every parameter <- uniform(-a, a): a = gain * sqrt(3 / fan_in) for weights, 0.1 for one-dimensional tensors; the stream is crc32 of the
    parameter's name, so the state_dict contract (joint_model.py's names) is what keys the values"""
    with torch.no_grad():
        for name, p in module.named_parameters():
            u = hashed_uniform(p.numel(), zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF, seed)
            fan_in = int(np.prod(p.shape[1:])) if p.dim() >= 2 else max(int(p.numel()), 1)
            bound = 0.1 if p.dim() == 1 else gain * math.sqrt(3.0 / fan_in)
            p.copy_(torch.from_numpy((2.0 * u - 1.0) * np.float32(bound)).view_as(p))
    return module


def synthetic_image(batch, side, seed=2):
    """(B, 1, S, S, S) ~ clip(N(0, 1), -1, 1): Box-Muller on two hashed streams"""
    n = batch * side ** 3
    u1 = np.maximum(hashed_uniform(n, 1001, seed), 1e-7).astype(np.float64)
    u2 = hashed_uniform(n, 1002, seed).astype(np.float64)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(np.clip(g, -1, 1).astype(np.float32)).view(batch, 1, side, side, side)


def synthetic_label(batch, side, seed=3):
    """(B, 1, S, S, S) float labels in {0, 1}: a centred ellipsoid with a hashed ragged rim"""
    u = hashed_uniform(batch * side ** 3, 2001, seed).reshape(batch, side, side, side)
    ax = (np.arange(side, dtype=np.float32) + 0.5) / side - 0.5
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    r = (z / 0.30) ** 2 + (y / 0.22) ** 2 + (x / 0.36) ** 2
    lab = ((r[None] + 0.35 * (u - 0.5)) < 1.0).astype(np.float32)
    return torch.from_numpy(lab).view(batch, 1, side, side, side)
