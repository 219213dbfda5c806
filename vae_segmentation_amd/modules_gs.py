"""The `*_GS` family of the reference's joint_model.py on the libvaeseg kernels: GSNorm3d (:17-33), DoubleConv_GS / Up_GS / Down_GS /
Conv_GS (:54-99), the weight-normalising convolutions GSConv3d / GSConvTranspose3d / SConv3d (:140-202) and Segmentation_GS (:307-346).

Nothing in the reference instantiates these classes (main_source.py / main_target.py build Segmentation / VAE / Encoder / Fusion only), so
they are served by the general, unfused passes: conv (live bias) -> ops.NormAct("none") for conv + ReLU, ops.GSNorm, ops.UpsampleTrilinear,
ops.Softmax2; the channel concatenation and the tiny weight transforms are ordinary torch ops (autograd differentiates them).  Class names,
constructor / forward signatures and state_dict keys are the reference's.  GPU only."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .modules import Act, _DEFAULT_DTYPE, _activation, _as_act, _as_tensor, _check_n_class


def _conv_act(conv, actm, a):
    """Conv3d(3, padding=1) -> ReLU / Softplus, no normalisation (joint_model.py:58-63,94-96)"""
    y, ys = ops.ConvK3.apply(a.raw, a.stats, conv.weight, conv.bias, True)
    act = ops.VS_ACT_SOFTPLUS if isinstance(actm, nn.Softplus) else ops.VS_ACT_RELU
    return Act(ops.NormAct.apply(y, ys, None, None, "none", act, conv.weight.shape[0]), None)


class GSNorm3d(nn.Module):
    """joint_model.py:17-33 — x[:, group] / (sum over the group's channels + 1e-4), groups of out_ch // num_group consecutive channels"""

    def __init__(self, out_ch, num_group=1):
        super().__init__()
        self.out_ch, self.num_group = out_ch, num_group
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        if a.stats is not None:
            a = Act(ops.Materialize.apply(a.raw, a.stats, None, None), None)
        if a.raw.shape[-1] != self.out_ch:
            raise ValueError("GSNorm3d(%d) got %d stored channels" % (self.out_ch, a.raw.shape[-1]))
        out = Act(ops.GSNorm.apply(a.raw, self.num_group), None)
        return _as_tensor(out, self.out_ch) if wrapped else out


class DoubleConv_GS(nn.Module):
    """joint_model.py:54-66 — (conv3x3x3 -> act) twice, Sequential indices 0..3"""

    def __init__(self, in_ch, out_ch, num_group=1, soft=False):
        super().__init__()
        activation = _activation(soft, False)
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, 3, padding=1), activation, nn.Conv3d(out_ch, out_ch, 3, padding=1), activation)
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        for i in (0, 2):
            a = _conv_act(self.conv[i], self.conv[i + 1], a)
        return _as_tensor(a, self.out_ch) if wrapped else a


class Up_GS(nn.Module):
    """joint_model.py:67-77 — Upsample(2, trilinear) -> DoubleConv_GS"""

    def __init__(self, in_ch, out_ch, num_group=1, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Upsample(scale_factor=2, mode="trilinear"), DoubleConv_GS(in_ch, out_ch, num_group, soft=False))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        if a.stats is not None:
            a = Act(ops.Materialize.apply(a.raw, a.stats, None, None), None)
        a = self.conv[1](Act(ops.UpsampleTrilinear.apply(a.raw, 2), None))
        return _as_tensor(a, self.out_ch) if wrapped else a


class Down_GS(nn.Module):
    """joint_model.py:78-88 — Conv3d(in, in, 2, stride 2) -> DoubleConv_GS"""

    def __init__(self, in_ch, out_ch, num_group=1, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        if tuple(kernal_size) != (2, 2, 2) or tuple(stride) != (2, 2, 2):
            raise NotImplementedError("native strided conv is written for kernel 2, stride 2")
        self.conv = nn.Sequential(nn.Conv3d(in_ch, in_ch, kernal_size, stride=stride, padding=0), DoubleConv_GS(in_ch, out_ch, num_group, soft=False))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        c = self.conv[0]
        a = self.conv[1](Act(ops.ConvK2S2.apply(a.raw, a.stats, c.weight, c.bias), None))
        return _as_tensor(a, self.out_ch) if wrapped else a


class Conv_GS(nn.Module):
    """joint_model.py:89-99 — conv3x3x3 -> act"""

    def __init__(self, in_ch, out_ch, num_group=1, activation=True, norm=True, soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, 3, padding=1), _activation(soft, True))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        a = _conv_act(self.conv[0], self.conv[1], a)
        return _as_tensor(a, self.out_ch) if wrapped else a


# ---- convolutions that transform their weight first (joint_model.py:140-202) ----------------------------------------------------------
def _group_normalised(weight, num_group):
    """|w| divided by its sum over each group of in_channels // num_group input channels (joint_model.py:154-161,183-189)"""
    w = torch.abs(weight)
    interval = weight.shape[1] // num_group
    parts = [w[:, i:i + interval] / torch.sum(w[:, i:i + interval], 1, keepdim=True) for i in range(0, interval * num_group, interval)]
    return torch.cat(parts, 1)


def _native_conv(x, weight, bias, module, dtype, transposed=False):
    """planar (N,C,D,H,W) -> planar through the native conv that matches the module's geometry; weight = the transformed (non-leaf) tensor"""
    k, st, pd = tuple(module.kernel_size), tuple(module.stride), tuple(module.padding)
    if tuple(module.dilation) != (1, 1, 1) or module.groups != 1:
        raise NotImplementedError("native convs: dilation 1, groups 1")
    a, _ = _as_act(x, dtype)
    if transposed:
        if k != (2, 2, 2) or st != (2, 2, 2) or pd != (0, 0, 0) or tuple(module.output_padding) != (0, 0, 0):
            raise NotImplementedError("native transposed conv: kernel 2, stride 2, no padding")
        return _as_tensor(Act(ops.ConvT2S2.apply(a.raw, None, weight, bias), None), weight.shape[1])
    if k == (3, 3, 3) and st == (1, 1, 1) and pd == (1, 1, 1):
        y, _ = ops.ConvK3.apply(a.raw, None, weight, bias, True)
        return _as_tensor(Act(y, None), weight.shape[0])
    if k == (2, 2, 2) and st == (2, 2, 2) and pd == (0, 0, 0):
        return _as_tensor(Act(ops.ConvK2S2.apply(a.raw, None, weight, bias), None), weight.shape[0])
    raise NotImplementedError("native convs: 3x3x3 / stride 1 / padding 1 or 2x2x2 / stride 2 / padding 0")


class GSConv3d(nn.Conv3d):
    """joint_model.py:140-163"""

    def __init__(self, in_channels, out_channels, kernel_size, num_group=1, stride=1, padding=0, dilation=1, groups=1, bias=True, if_sub=None,
                 trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if not trainable:
            self.weight.requires_grad = False
        self.num_group = num_group
        self.interval = self.in_channels // self.num_group
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        return _native_conv(x, _group_normalised(self.weight, self.num_group), self.bias, self, self.kernel_dtype)


class GSConvTranspose3d(nn.ConvTranspose3d):
    """joint_model.py:166-190"""

    def __init__(self, in_channels, out_channels, kernel_size, num_group=1, stride=1, padding=0, dilation=1, output_padding=0, groups=1, bias=False,
                 if_sub=None, trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias, dilation)
        if not trainable:
            self.weight.requires_grad = False
        self.num_group = num_group
        self.interval = self.in_channels // self.num_group
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x, output_size=None):
        if output_size is not None:
            raise NotImplementedError("native transposed conv: output_size is implied by kernel 2 / stride 2")
        return _native_conv(x, _group_normalised(self.weight, self.num_group), self.bias, self, self.kernel_dtype, transposed=True)


class SConv3d(nn.Conv3d):
    """joint_model.py:191-202 — the kernel's spatial mean is subtracted from every filter"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, if_sub=None, trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if not trainable:
            self.weight.requires_grad = False
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        return _native_conv(x, self.weight - self.weight.mean([2, 3, 4], keepdim=True), self.bias, self, self.kernel_dtype)


class Segmentation_GS(nn.Module):
    """joint_model.py:307-346 — four-level encoder without normalisation layers; every level is group-normalised (GSNorm3d), brought to
    full resolution (trilinear), concatenated (8+16+32+64 = 120 channels), then conv3x3x3 -> ReLU -> conv1x1x1 -> Softmax."""

    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=[8, 16, 32, 64, 128, 256]):
        super().__init__()
        _check_n_class(n_class)
        f = list(n_fmaps)
        self.in_block = Conv_GS(n_channels, f[0], num_group=2, soft=False)
        self.down1 = Down_GS(f[0], f[1], num_group=2, soft=False)
        self.down2 = Down_GS(f[1], f[2], num_group=2, soft=False)
        self.down3 = Down_GS(f[2], f[3], num_group=4, soft=False)
        self.norm1 = GSNorm3d(f[0], num_group=2)
        self.norm2 = GSNorm3d(f[1], num_group=4)
        self.norm3 = GSNorm3d(f[2], num_group=8)
        self.norm4 = GSNorm3d(f[3], num_group=8)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear")
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear")
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear")
        self.out_block1 = Conv_GS(f[0] + f[1] + f[2] + f[3], 32, soft=False)
        self.out_block2 = nn.Conv3d(32, n_class, 1, padding=0)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, data_dict, in_key, out_key):
        x = data_dict[in_key]
        ops._require_cuda(x)
        if any(s % 8 for s in x.shape[2:]):
            raise ValueError("Segmentation_GS needs spatial sizes that are multiples of 8, got %s" % (tuple(x.shape[2:]),))
        ops.stats_arena_begin(x.device)
        x1 = self.in_block(Act(ops.PackPlanar.apply(x, self.kernel_dtype), None))
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        parts = [self.norm1(x1).raw,
                 ops.UpsampleTrilinear.apply(self.norm2(x2).raw, 2),
                 ops.UpsampleTrilinear.apply(self.norm3(x3).raw, 4),
                 ops.UpsampleTrilinear.apply(self.norm4(x4).raw, 8)]
        cat = torch.cat(parts, dim=-1)                                         # channels-last: the reference's torch.cat(dim=1)
        pad = ops.cpad(cat.shape[-1]) - cat.shape[-1]
        if pad:
            cat = F.pad(cat, (0, pad))                                         # 120 -> 128 stored channels (zero weights meet them)
        h = self.out_block1(Act(cat, None))
        # the 1x1x1 out_block2 as the centre tap of a 3x3x3 kernel whose other 26 taps are zero: the same sums, on the existing conv
        w3 = F.pad(self.out_block2.weight, (1, 1, 1, 1, 1, 1))
        logits, _ = ops.ConvK3.apply(h.raw, None, w3, self.out_block2.bias, True)
        data_dict[out_key] = ops.Softmax2.apply(logits, self.out_block2.weight.shape[0])
        return data_dict
