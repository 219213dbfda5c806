"""CPU: the data-pipeline oracle (oracle/data_cpu.py).  Its third-party steps (skimage resize, batchgenerators augment_spatial) are
restated on scipy.ndimage and cannot be pinned against those libraries here (not installed: "parity unpinned" in its header); what can
be checked are known answers of the reference's own numpy steps and the invariants any correct restatement must have."""
import numpy as np
import pytest

from oracle import data_cpu as D


def test_crop_geometry_known_answers():
    """utils/utils.py:345-378 by hand: label box z 10..29, y 20..59, x 5..14 in a 64 x 80 x 40 volume"""
    lab = np.zeros((64, 80, 40), np.float32)
    lab[10:30, 20:60, 5:15] = 1
    centre, L, pad = D.crop_box(lab)
    assert centre.tolist() == [(29 + 10) // 2, (59 + 20) // 2, (14 + 5) // 2] == [19, 39, 9]
    assert L == 39 and pad == 3                                   # largest extent 59 - 20, int(3.9)
    img = np.arange(lab.size, dtype=np.float32).reshape(lab.shape)
    cube = D.crop_pad_cube(img, centre, L, pad)
    assert cube.shape == (45, 45, 45)                             # L + 2 pad
    # z: [19-19-3, 19+19+3) = [-3, 41) -> clipped [0, 41): 41 rows, 4 missing -> 2 before, 2 after
    assert cube[1].max() == 0 and cube[2].max() > 0 and cube[42].max() > 0 and cube[43].max() == 0
    # y: [39-19-3, 39+19+3) = [17, 61): 44 rows, 1 missing -> 0 before, 1 after;  x: [-13, 31) -> [0, 31): 31 columns, 14 missing -> 7 / 7
    assert cube[10, 10, 6] == 0 and cube[10, 10, 7] == img[8, 27, 0] and cube[10, 10, 37] == img[8, 27, 30] and cube[10, 10, 38] == 0
    assert cube[10, 43, 10] == img[8, 60, 3] and cube[10, 44, 10] == 0
    empty_centre, empty_L, empty_pad = D.crop_box(np.zeros((8, 8, 8), np.float32))
    assert empty_centre.tolist() == [64, 64, 64] and empty_L == 32 and empty_pad == 3      # utils/utils.py:355-358


def test_relabel_and_intensity_steps():
    merge = np.zeros((4, 4, 4, 2), np.float32)
    merge[..., 1] = np.arange(64).reshape(4, 4, 4) % 5
    img, lab = D.load_merge(merge, [[[1, 3], 1], [2, 7]])
    assert set(np.unique(lab)) == {0.0, 1.0, 7.0} and (lab == 1).sum() == ((merge[..., 1] == 1) | (merge[..., 1] == 3)).sum()
    x = np.array([-1000.0, -200.0, 100.0, 400.0, 3000.0])
    assert np.allclose(D.center_intensities(D.clip(x)), [-1.0, -1.0, 0.0, 1.0, 1.0])          # main_source.py:209-210


def test_resize_invariants():
    rng = np.random.RandomState(0)
    img = rng.randn(20, 24, 28).astype(np.float32)
    assert np.array_equal(D.skimage_resize(img, img.shape), img)                       # same shape: zoom 1, no anti-aliasing
    for out in ((40, 40, 40), (10, 12, 14)):
        r = D.skimage_resize(img, out)
        assert r.shape == out and r.dtype == np.float32 and r.min() >= img.min() and r.max() <= img.max()          # clip=True
    assert np.allclose(D.skimage_resize(np.full((9, 9, 9), 3.5, np.float32), (16, 16, 16)), 3.5)
    lab = (rng.rand(20, 24, 28) > 0.5).astype(np.float32) * 2
    r0 = D.skimage_resize(lab, (32, 32, 32), order=0, anti_aliasing=False)
    assert set(np.unique(r0)) <= {0.0, 2.0}
    assert np.array_equal(D.skimage_resize(lab, (40, 48, 56), order=0, anti_aliasing=False)[::2, ::2, ::2], lab)    # 2x nearest repeats voxels


def test_spatial_transform_invariants():
    rng = np.random.RandomState(1)
    img = rng.randn(12, 12, 12).astype(np.float32)
    lab = (rng.rand(12, 12, 12) > 0.5).astype(np.float32)
    mid = ((12 - 1) / 2.0,) * 3
    i0, l0 = D.spatial_transform(img, lab, (12, 12, 12), (0.0, 0.0, 0.0), 1.0, mid)
    assert np.allclose(i0, img, atol=1e-5) and np.array_equal(l0, lab)                  # identity: a cubic spline interpolates its samples
    r = D.rotation_matrix(0.1, -0.2, 0.15)
    assert np.allclose(r @ r.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(r), 1.0)
    # a quarter turn about the first axis maps the grid onto itself: the result is a rot90 of the (y, x) planes
    iq, _ = D.spatial_transform(img, lab, (12, 12, 12), (np.pi / 2, 0.0, 0.0), 1.0, mid)
    assert min(np.abs(iq - np.rot90(img, k, axes=(1, 2))).max() for k in (1, 3)) < 1e-4
    # zooming out by 2 around the centre leaves a constant border
    iz, lz = D.spatial_transform(img, lab, (12, 12, 12), (0.0, 0.0, 0.0), 2.0, mid)
    assert iz[0, 0, 0] == -1024.0 and lz[0, 0, 0] == 0.0


def test_draw_order_and_ranges():
    p = D.draw_spatial_params(np.random.RandomState(3), (128, 128, 128), (128, 128, 128), [59] * 3)
    q = D.draw_spatial_params(np.random.RandomState(3), (128, 128, 128), (128, 128, 128), [59] * 3)
    assert p == q and p["modified"]
    assert all(-0.2 <= a <= 0.2 for a in p["angles"]) and 0.85 <= p["scale"] <= 1.15 and all(59 <= c <= 69 for c in p["centre"])
    scales = [D.draw_spatial_params(np.random.RandomState(s), (64,) * 3, (64,) * 3, [27] * 3)["scale"] for s in range(200)]
    assert 0.3 < np.mean(np.array(scales) < 1.0) < 0.7                                  # half the draws shrink, half enlarge (augment_spatial's coin)
