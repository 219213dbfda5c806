"""CPU oracle for the training data pipeline.  TEST INFRASTRUCTURE ONLY — numpy + scipy.

What it follows (all under /root/reference):
  * ``load_merge``          utils/utils.py:220-276  NumpyLoader_Multi_merge: channel 0 of the merge array is the image, channel 1 the label
                            map, relabelled through ``mask_index`` pairs ([source labels], target)
  * ``crop_resize``         utils/utils.py:326-383  CropResize (the training branch): bounding box of label > 0, centre (max+min)//2,
                            L = largest extent, pad = int(0.1 L); crop [c - L//2 - pad + shift, c + L//2 + pad + shift) clipped to the
                            volume, zero-padded to a cube of side L + 2 pad (split int(diff/2) / rest); image resized with skimage's
                            ``resize`` defaults, label with order 0 / no anti-aliasing
  * ``spatial_transform``   utils/utils.py:927-968 MySpatialTransform = batchgenerators' augment_spatial as configured by
                            main_source.py:196-205: rotation about x, y, z, isotropic scale, random crop centre, cubic-spline image /
                            nearest label interpolation, constant borders (-1024 / 0), no elastic deformation
  * ``clip`` / ``center_intensities``   utils/utils.py:508-533, 575-618 (main_source.py:209-210: clip to [-200, 400], (x - 100) / 300)

PARITY UNPINNED for the two third-party steps: ``skimage.transform.resize`` and ``batchgenerators...augment_spatial`` are imported by
the reference (utils/utils.py:4,19) but are not installed here (no requirements file pins their versions), and the reference holds
no fixtures for them.  They are restated from their published algorithms on top of the scipy.ndimage primitives both libraries call
(gaussian_filter + zoom(grid_mode=True) for resize as of scikit-image 0.19+; map_coordinates for augment_spatial), which ARE the
arithmetic; what cannot be checked here is only their glue (default arguments, the order of random draws).  The numpy-only steps
(crop geometry, relabelling, clip, centring) restate the reference's own code.
"""
import numpy as np
from scipy import ndimage as ndi


def load_merge(merge, mask_index=None, dtype=np.float32):
    """merge: (D, H, W, >=2) array -> (image, label)"""
    img = merge[..., 0].astype(dtype)
    if mask_index is None:
        return img, merge[..., 1].astype(dtype)
    lab = np.zeros_like(merge[..., 1])
    for sources, target in mask_index:
        for s in (sources if isinstance(sources, (list, tuple)) else [sources]):
            lab[merge[..., 1] == s] = target
    return img, lab.astype(dtype)


def crop_box(label, shift=0):
    """-> (centre[3], L, pad) of CropResize; an empty label gives the reference's fallback centre (64, 64, 64), L = 32"""
    idx = np.array(np.where(label > 0)).T
    if idx.shape[0] > 0:
        bmax, bmin = idx.max(0), idx.min(0)
        centre = (bmax + bmin) // 2
        L = int(np.max(bmax - bmin))
    else:
        centre, L = np.array([64, 64, 64]), 32
    return centre, L, int(L * 0.1)


def crop_pad_cube(vol, centre, L, pad, shift=0):
    """the clipped crop around `centre`, zero-padded to a cube of side L + 2 pad (utils/utils.py:364-378)"""
    lo = [max(int(centre[d]) - L // 2 - pad + shift, 0) for d in range(3)]
    hi = [min(int(centre[d]) + L // 2 + pad + shift, vol.shape[d]) for d in range(3)]
    out = vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    diff = [L + pad * 2 - s for s in out.shape]
    return np.pad(out, [(int(d / 2), d - int(d / 2)) for d in diff])


def skimage_resize(image, output_shape, order=1, anti_aliasing=None):
    """skimage.transform.resize(image, output_shape, order=order, anti_aliasing=anti_aliasing) with its other defaults (mode='reflect',
    cval=0, clip=True, preserve_range=False) for a float image, restated on scipy.ndimage as scikit-image >= 0.19 implements it."""
    image = np.asarray(image)
    in_shape = np.array(image.shape, dtype=np.float64)
    factors = in_shape / np.array(output_shape, dtype=np.float64)
    if anti_aliasing is None:
        anti_aliasing = bool(np.any(np.array(output_shape) < np.array(image.shape)))
    filtered = image.astype(np.float64) if order > 0 else image
    if anti_aliasing:
        sigma = np.maximum(0, (factors - 1) / 2)
        filtered = ndi.gaussian_filter(filtered, sigma, cval=0, mode="mirror")
    out = ndi.zoom(filtered, 1.0 / factors, order=order, mode="mirror", cval=0, grid_mode=True)
    if order > 0:
        out = np.clip(out, image.min(), image.max())
    return out.astype(image.dtype)


def crop_resize(img, label, output_size, shift=0, pred=None):
    """pred (a coarse prediction, utils/utils.py:345-358): when given, the box comes from pred > 0 instead of the label, pred itself is cropped
    (without the shift), padded and resized with order 0, and a third array is returned"""
    centre, L, pad = crop_box(label if pred is None else pred)
    img_c, lab_c = crop_pad_cube(img, centre, L, pad, shift), crop_pad_cube(label, centre, L, pad, shift)
    out = (skimage_resize(img_c, output_size), skimage_resize(lab_c, output_size, order=0, anti_aliasing=False))
    if pred is not None:
        out += (skimage_resize(crop_pad_cube(pred, centre, L, pad, 0), output_size, order=0, anti_aliasing=False),)
    return out


# ---- batchgenerators.augmentations.spatial_transformations.augment_spatial, as MySpatialTransform configures it -------------------------
def rotation_matrix(ax, ay, az):
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return np.identity(3) @ rx @ ry @ rz


def draw_spatial_params(rng, shape, patch_size, dist_from_border, scale=(0.85, 1.15), angle=(-0.2, 0.2), p_rot=1.0, p_scale=1.0):
    """the random draws of one sample, in augment_spatial's order (elastic deformation off, p_el_per_sample = 0: its uniform() draw is
    still taken because `do_elastic_deform and uniform() < p` short-circuits on do_elastic_deform=False: NOT taken) -> dict"""
    out = {"angles": (0.0, 0.0, 0.0), "scale": 1.0, "modified": False}
    if rng.uniform() < p_rot:
        out["angles"] = tuple(rng.uniform(angle[0], angle[1]) if rng.uniform() <= 1.0 else 0.0 for _ in range(3))
        out["modified"] = True
    if rng.uniform() < p_scale:
        if rng.random_sample() < 0.5 and scale[0] < 1:
            out["scale"] = rng.uniform(scale[0], 1)
        else:
            out["scale"] = rng.uniform(max(scale[0], 1), scale[1])
        out["modified"] = True
    out["centre"] = tuple(rng.uniform(dist_from_border[d], shape[d] - dist_from_border[d]) for d in range(3))
    return out


def spatial_coords(patch_size, angles, scale, centre):
    """(3, *patch_size) sampling coordinates: zero-centred mesh, rotated (row vectors times R), scaled, moved to `centre`"""
    mesh = np.array(np.meshgrid(*[np.arange(s) for s in patch_size], indexing="ij")).astype(float)
    for d in range(3):
        mesh[d] -= (patch_size[d] - 1) / 2.0
    c = (mesh.reshape(3, -1).T @ rotation_matrix(*angles)).T.reshape(mesh.shape) * scale
    for d in range(3):
        c[d] += centre[d]
    return c


def spatial_transform(img, label, patch_size, angles, scale, centre, cval_img=-1024.0, cval_seg=0.0):
    c = spatial_coords(patch_size, angles, scale, centre)
    out_i = ndi.map_coordinates(img.astype(float), c, order=3, mode="constant", cval=cval_img).astype(img.dtype)
    out_l = ndi.map_coordinates(label.astype(float), c, order=0, mode="constant", cval=cval_seg).astype(label.dtype)
    return out_i, out_l


def clip(x, lo=-200.0, hi=400.0):
    return np.clip(x, lo, hi)


def center_intensities(x, subtrahend=100.0, divisor=300.0):
    return (x - subtrahend) / divisor


def train_sample(merge, patch_size, params, mask_index=None):
    """one training sample through main_source.py:191-211 (augmentation on) given the spatial parameters -> (image, label), fp32"""
    img, lab = load_merge(merge, mask_index)
    img, lab = crop_resize(img, lab, patch_size)
    if params is not None:
        img, lab = spatial_transform(img, lab, patch_size, params["angles"], params["scale"], params["centre"])
    return center_intensities(clip(img)).astype(np.float32), lab.astype(np.float32)
