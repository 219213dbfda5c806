"""The training data pipeline of the reference (main_source.py:191-211) on the device: the transform classes of utils/utils.py with the
same names, constructor arguments and dict-in / dict-out protocol, operating on CUDA tensors through libvaeseg's vs_data_* kernels.

    NumpyLoader_Multi_merge   utils/utils.py:220-276   (the relabelling; file I/O stays with the caller: hand it the merge array)
    CropResize                utils/utils.py:326-383   (bounding box of the label or of a coarse prediction, cube crop + zero pad, resize)
    MySpatialTransform        utils/utils.py:927-968   (batchgenerators augment_spatial: rotation, scale, random crop; elastic deformation,
                                                        which main_source.py:198 switches off, is not implemented)
    Clip, CenterIntensities   utils/utils.py:508-533, 575-618

The reference runs this chain on 16 CPU workers per loader (skimage resize + scipy map_coordinates of 128^3 volumes: seconds per
sample); here a sample costs a handful of kernel launches.  One host synchronisation per sample remains: the crop cube's side depends
on the label's bounding box, so six integers come back before the crop buffers are sized.  Random parameters are drawn on the host
from numpy's RandomState in augment_spatial's order (a dozen scalars per sample).  Arithmetic parity: tests/test_gpu_data.py against
oracle/data_cpu.py (scipy.ndimage)."""
import ctypes

import numpy as np
import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _vol(t):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.is_contiguous()):
        raise TypeError("expected a contiguous CUDA float32 volume (D, H, W), got %s %s on %s" % (tuple(t.shape), t.dtype, t.device))
    return t


def _ints(vals):
    return (ctypes.c_int * 3)(*[int(v) for v in vals])


def relabel(label_map, mask_index):
    """NumpyLoader_Multi_merge's mask_index handling (utils/utils.py:253-262): pairs ([source labels], target)"""
    _vol(label_map)
    src, dst = [], []
    for sources, target in mask_index:
        for s in (sources if isinstance(sources, (list, tuple)) else [sources]):
            src.append(float(s)); dst.append(float(target))
    out = torch.empty_like(label_map)
    check(lib.vs_data_relabel(label_map.data_ptr(), out.data_ptr(), label_map.numel(), (ctypes.c_float * len(src))(*src),
                              (ctypes.c_float * len(dst))(*dst), len(src), _stream()), "data_relabel")
    return out


def bounding_box(label):
    """-> (min[3], max[3]) of label > 0 as numpy ints, or None when the label is empty (one host synchronisation)"""
    _vol(label)
    box = torch.empty(6, dtype=torch.int32, device=label.device)
    check(lib.vs_data_bbox(label.data_ptr(), *label.shape, box.data_ptr(), _stream()), "data_bbox")
    b = box.cpu().numpy()
    return None if b[3] < 0 else (b[:3].astype(np.int64), b[3:].astype(np.int64))


def crop_pad_cube(vol, centre, L, pad, shift=0):
    _vol(vol)
    lo = [max(int(centre[d]) - L // 2 - pad + shift, 0) for d in range(3)]
    hi = [min(int(centre[d]) + L // 2 + pad + shift, vol.shape[d]) for d in range(3)]
    side = L + 2 * pad
    off = [int((side - (hi[d] - lo[d])) / 2) for d in range(3)]
    out = torch.empty((side, side, side), dtype=torch.float32, device=vol.device)
    check(lib.vs_data_crop_pad(vol.data_ptr(), out.data_ptr(), *vol.shape, side, side, side, _ints(lo), _ints(hi), _ints(off), _stream()), "data_crop_pad")
    return out


def resize(vol, output_size, order=1, anti_aliasing=None):
    """skimage.transform.resize(vol, output_size, order=order, anti_aliasing=anti_aliasing) with its other defaults, for a float volume
    (see oracle/data_cpu.py:skimage_resize for the restated algorithm)"""
    _vol(vol)
    out_shape = tuple(int(s) for s in output_size)
    if anti_aliasing is None:
        anti_aliasing = any(o < i for o, i in zip(out_shape, vol.shape))
    lo, hi = 1.0, 0.0                                    # lo > hi: no clipping
    if order > 0:
        mm = torch.empty(2, dtype=torch.float32, device=vol.device)
        check(lib.vs_data_minmax(vol.data_ptr(), vol.numel(), mm.data_ptr(), _stream()), "data_minmax")
        lo, hi = [float(v) for v in mm.cpu()]            # clip=True: to the input's range
    src = vol
    if anti_aliasing and order > 0:
        for axis in range(3):
            sigma = max(0.0, (vol.shape[axis] / out_shape[axis] - 1.0) / 2.0)
            if sigma > 0.0:
                nxt = torch.empty_like(src)
                check(lib.vs_data_gaussian_axis(src.data_ptr(), nxt.data_ptr(), *src.shape, axis, sigma, _stream()), "data_gaussian_axis")
                src = nxt
    out = torch.empty(out_shape, dtype=torch.float32, device=vol.device)
    check(lib.vs_data_zoom(src.data_ptr(), out.data_ptr(), *src.shape, *out_shape, 0 if order == 0 else 1, lo, hi, _stream()), "data_zoom")
    return out


class BaseTransform:
    def __init__(self, fields):
        self.fields = [fields] if isinstance(fields, str) else list(fields)


class CropResize(BaseTransform):
    """utils/utils.py:326-383 (fields f, f + '_pancreas' and, when present, the coarse prediction f + '_pancreas_pred' that then defines the box)"""

    def __init__(self, fields, output_size, pad=32, shift=0):
        super().__init__(fields)
        self.output_size, self.pad, self.shift = output_size, pad, shift

    def __call__(self, data_dict):
        for f in self.fields:
            if data_dict.get(f) is None:
                continue
            img, label = data_dict[f], data_dict[f + "_pancreas"]
            pred = data_dict.get(f + "_pancreas_pred")
            if isinstance(pred, torch.Tensor):               # utils/utils.py:345-358: the box of a coarse prediction (which the reference assumes non-empty)
                box = bounding_box(pred)
                if box is None:
                    raise ValueError("CropResize: empty %s_pancreas_pred (the reference fails on np.max of an empty index array here)" % f)
            else:
                box = bounding_box(label)
            if box is not None:
                centre, L = (box[1] + box[0]) // 2, int(np.max(box[1] - box[0]))
            else:
                centre, L = np.array([64, 64, 64]), 32
            pad = int(L * 0.1)
            if isinstance(pred, torch.Tensor):
                data_dict[f + "_pancreas_pred"] = resize(crop_pad_cube(pred, centre, L, pad, 0), self.output_size, order=0, anti_aliasing=False)
            lab_c = crop_pad_cube(label, centre, L, pad, self.shift)
            data_dict["ori_shape"] = np.array(list(label.shape) + list(lab_c.shape))
            data_dict[f] = resize(crop_pad_cube(img, centre, L, pad, self.shift), self.output_size)
            data_dict[f + "_pancreas"] = resize(lab_c, self.output_size, order=0, anti_aliasing=False)
        return data_dict


def rotation_matrix(ax, ay, az):
    """batchgenerators' create_matrix_rotation_{x,y,z}_3d chained from the identity"""
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return np.identity(3) @ rx @ ry @ rz


def affine_resample(vol, patch_size, angles, scale, centre, order, cval):
    """one channel through augment_spatial's coordinate map: coords = (mesh - (P-1)/2) . R * scale + centre, then
    scipy.ndimage.map_coordinates(order, mode='constant', cval)"""
    _vol(vol)
    a = (scale * rotation_matrix(*angles).T).astype(np.float64).reshape(-1)          # row vectors times R == R^T times column vectors
    a9, c3 = (ctypes.c_double * 9)(*a), (ctypes.c_double * 3)(*[float(v) for v in centre])
    out = torch.empty(tuple(int(s) for s in patch_size), dtype=torch.float32, device=vol.device)
    if order == 3:
        coef = torch.empty(vol.shape, dtype=torch.float64, device=vol.device)
        check(lib.vs_data_spline3_prefilter(vol.data_ptr(), coef.data_ptr(), *vol.shape, _stream()), "data_spline3_prefilter")
        src = coef
    elif order == 0:
        src = vol
    else:
        raise NotImplementedError("native resampling: order 3 (image) or 0 (label), the reference's settings")
    check(lib.vs_data_affine_sample(src.data_ptr(), out.data_ptr(), *vol.shape, *out.shape, a9, c3, order, float(cval), _stream()), "data_affine_sample")
    return out


class MySpatialTransform:
    """utils/utils.py:927-968 with the arguments main_source.py:196-205 passes.  data_dict[data_key] / [label_key]: (B, C, D, H, W) CUDA
    tensors (the reference reshapes to [-1, 1, D, H, W] first).  `rng`: a numpy RandomState (default: the global one, as batchgenerators)."""

    def __init__(self, patch_size, patch_center_dist_from_border=30, do_elastic_deform=True, alpha=(0., 1000.), sigma=(10., 13.), do_rotation=True,
                 angle_x=(0, 2 * np.pi), angle_y=(0, 2 * np.pi), angle_z=(0, 2 * np.pi), do_scale=True, scale=(0.75, 1.25), border_mode_data="nearest",
                 border_cval_data=0, order_data=3, border_mode_seg="constant", border_cval_seg=0, order_seg=0, random_crop=True, data_key="data",
                 label_key="seg", p_el_per_sample=1, p_scale_per_sample=1, p_rot_per_sample=1, independent_scale_for_each_axis=False,
                 p_rot_per_axis: float = 1, rng=None):
        if do_elastic_deform and p_el_per_sample > 0:
            raise NotImplementedError("elastic deformation has no native kernel (main_source.py:198 passes do_elastic_deform=False)")
        if border_mode_data != "constant" or border_mode_seg != "constant" or independent_scale_for_each_axis:
            raise NotImplementedError("native resampling: constant borders, isotropic scale (main_source.py:196-205)")
        self.patch_size = patch_size
        self.dist = patch_center_dist_from_border if isinstance(patch_center_dist_from_border, (list, tuple, np.ndarray)) else 3 * [patch_center_dist_from_border]
        self.do_rotation, self.angle = do_rotation, (angle_x, angle_y, angle_z)
        self.do_scale, self.scale = do_scale, scale
        self.cval_data, self.cval_seg, self.order_data, self.order_seg = border_cval_data, border_cval_seg, order_data, order_seg
        self.random_crop, self.data_key, self.label_key = random_crop, data_key, label_key
        self.p_scale, self.p_rot, self.p_rot_axis = p_scale_per_sample, p_rot_per_sample, p_rot_per_axis
        self.rng = rng if rng is not None else np.random

    def draw(self, shape):
        """the random draws of one sample in augment_spatial's order -> (angles, scale, centre, modified)"""
        r = self.rng
        angles, sc, modified = [0.0, 0.0, 0.0], 1.0, False
        if self.do_rotation and r.uniform() < self.p_rot:
            for d in range(3):
                angles[d] = r.uniform(self.angle[d][0], self.angle[d][1]) if r.uniform() <= self.p_rot_axis else 0.0
            modified = True
        if self.do_scale and r.uniform() < self.p_scale:
            if r.random_sample() < 0.5 and self.scale[0] < 1:
                sc = r.uniform(self.scale[0], 1)
            else:
                sc = r.uniform(max(self.scale[0], 1), self.scale[1])
            modified = True
        if self.random_crop:
            centre = [r.uniform(self.dist[d], shape[d] - self.dist[d]) for d in range(3)]
        else:
            centre = [shape[d] / 2.0 - 0.5 for d in range(3)]
        return tuple(angles), sc, tuple(centre), modified

    def __call__(self, data_dict, params=None):
        data, seg = data_dict.get(self.data_key), data_dict.get(self.label_key)
        patch = tuple(self.patch_size) if self.patch_size is not None else tuple(data.shape[2:])
        out_d = torch.empty((data.shape[0], data.shape[1]) + patch, dtype=torch.float32, device=data.device)
        out_s = None if seg is None else torch.empty((seg.shape[0], seg.shape[1]) + patch, dtype=torch.float32, device=seg.device)
        for b in range(data.shape[0]):
            angles, sc, centre, modified = params[b] if params is not None else self.draw(data.shape[2:])
            if not modified and not self.random_crop and tuple(data.shape[2:]) == patch:
                out_d[b] = data[b]
                if seg is not None:
                    out_s[b] = seg[b]
                continue
            for c in range(data.shape[1]):
                out_d[b, c] = affine_resample(data[b, c].contiguous(), patch, angles, sc, centre, self.order_data, self.cval_data)
            if seg is not None:
                for c in range(seg.shape[1]):
                    out_s[b, c] = affine_resample(seg[b, c].contiguous(), patch, angles, sc, centre, self.order_seg, self.cval_seg)
        data_dict[self.data_key] = out_d
        if seg is not None:
            data_dict[self.label_key] = out_s
        return data_dict


class Clip(BaseTransform):
    """utils/utils.py:508-533"""

    def __init__(self, fields, new_min=0.0, new_max=1.0):
        super().__init__(fields)
        self._new_min, self._new_max = new_min, new_max

    def __call__(self, data_dict):
        for f in self.fields:
            if data_dict.get(f) is not None:
                x = data_dict[f].contiguous()
                check(lib.vs_data_clip_center(x.data_ptr(), x.numel(), float(self._new_min), float(self._new_max), 0.0, 1.0, _stream()), "data_clip_center")
                data_dict[f] = x
        return data_dict


class CenterIntensities(BaseTransform):
    """utils/utils.py:575-618 (scalar subtrahend / divisor, as main_source.py:210 passes)"""

    def __init__(self, fields, subtrahend, divisor=1.0):
        super().__init__(fields)
        if isinstance(subtrahend, (list, tuple, np.ndarray)) or isinstance(divisor, (list, tuple, np.ndarray)):
            raise NotImplementedError("per-channel subtrahend / divisor lists: the reference passes scalars (main_source.py:210)")
        self.subtrahend, self.divisor = subtrahend, divisor

    def __call__(self, data_dict):
        for f in self.fields:
            if data_dict.get(f) is not None:
                x = data_dict[f].contiguous()
                check(lib.vs_data_clip_center(x.data_ptr(), x.numel(), -3.0e38, 3.0e38, float(self.subtrahend), float(self.divisor), _stream()), "data_clip_center")
                data_dict[f] = x
        return data_dict


def train_sample(merge, patch_size, mask_index=None, transform=None, params=None, field="venous", shift=0):
    """One training sample through main_source.py:191-211 on the device: merge (D, H, W, >= 2) CUDA float32 tensor (what
    NumpyLoader_Multi_merge loads) -> (image (1, 1, P, P, P), label (1, 1, P, P, P)).  `transform`: a MySpatialTransform (None: no
    augmentation, --no_aug); `params`: its per-sample (angles, scale, centre, modified) instead of random draws."""
    img = merge[..., 0].contiguous()
    lab = merge[..., 1].contiguous()
    if mask_index is not None:
        lab = relabel(lab, mask_index)
    d = {field: img, field + "_pancreas": lab}
    d = CropResize([field], patch_size, shift=shift)(d)                 # --shift: main_target.py:81,204 (training crops only)
    d[field], d[field + "_pancreas"] = d[field][None, None], d[field + "_pancreas"][None, None]
    if transform is not None:
        d = transform(d, params=None if params is None else [params])
    d = Clip([field], new_min=-200, new_max=400)(d)
    d = CenterIntensities([field], subtrahend=100, divisor=300)(d)
    return d[field], d[field + "_pancreas"]
