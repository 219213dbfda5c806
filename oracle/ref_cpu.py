"""CPU oracle for the 3D VAE + segmentation training path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement (stock ``nn.Conv3d`` / ``nn.ConvTranspose3d`` /
``nn.InstanceNorm3d`` / ``nn.Linear`` on the CPU) of the hot path of yyNoBug/VAE_segmentation.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``vae_segmentation_amd``) never does and has no CPU fallback.

Parity pin: the reference holds no tests / golden vectors of its own, so this
oracle is pinned against outputs of the *reference itself* run in the build container:
``oracle/make_golden.py`` imports ``/root/reference/joint_model.py`` and
``/root/reference/utils/evaluation.py`` unmodified and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those fixtures.

What each piece follows in the reference (file:line, all under /root/reference):
  * ``norm_layer``                joint_model.py:9-15   (norm_type 1 -> InstanceNorm3d, 2 -> BatchNorm3d)
  * ``Conv``                      joint_model.py:101-112
  * ``DoubleConv``                joint_model.py:35-52  (three conv/norm/act triples)
  * ``Up`` / ``Down``             joint_model.py:114-136
  * ``VAE``                       joint_model.py:204-272 (generalised: ``spatial`` replaces the hard-wired 128)
  * ``Segmentation``              joint_model.py:349-390
  * ``Joint``                     joint_model.py:438-452
  * ``dice_scores`` / ``avg_dsc`` utils/evaluation.py:48-80 (eps 1e-6) and main_source.py:150-182 (eps 1e-4)
  * ``KLloss``                    utils/evaluation.py:42-45
  * ``binarize`` / ``confident_binarize``  utils/evaluation.py:9-18
  * ``avg_ce``                    utils/evaluation.py:29-39
  * ``one_hot``                   main_source.py:449-451
  * ``joint_train_losses``        main_source.py:469-471
  * ``vae_train_losses``          main_source.py:393-413
  * ``seg_train_losses``          main_source.py:440-441
  * ``domain_adaptation_losses``  main_target.py:531-596
"""
import math
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

FMAPS = (8, 16, 32, 64, 128, 256)


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def norm_layer(norm_type, channels):
    """/root/reference/joint_model.py:9-15.  norm_type 3 = GSNorm3d with num_group 1: the reference's Conv / DoubleConv never forward their
    num_group argument (joint_model.py:41,44,47,107)."""
    if norm_type == 1:
        return nn.InstanceNorm3d(channels)
    if norm_type == 2:
        return nn.BatchNorm3d(channels, momentum=0.1)
    if norm_type == 3:
        return GSNorm3d(channels, num_group=1)
    raise ValueError("norm_type must be 1 (InstanceNorm3d), 2 (BatchNorm3d) or 3 (GSNorm3d)")


def _act(soft, inplace):
    return nn.Softplus() if soft else nn.ReLU(inplace=inplace)


def _cna(cin, cout, norm_type, soft, inplace):
    """conv3x3x3 -> norm -> activation, as three Sequential entries."""
    return [nn.Conv3d(cin, cout, 3, padding=1), norm_layer(norm_type, cout), _act(soft, inplace)]


class Conv(nn.Module):
    def __init__(self, in_ch, out_ch, norm_type=2, num_group=1, activation=True, norm=True, soft=False):
        super().__init__()
        self.conv = nn.Sequential(*_cna(in_ch, out_ch, norm_type, soft, True))

    def forward(self, x):
        return self.conv(x)


class DoubleConv(nn.Module):
    def __init__(self, in_ch, out_ch, norm_type=2, soft=False):
        super().__init__()
        layers = []
        for cin in (in_ch, out_ch, out_ch):
            layers += _cna(cin, out_ch, norm_type, soft, False)
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return self.conv(x)


class Up(nn.Module):
    def __init__(self, in_ch, out_ch, norm_type=2, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.ConvTranspose3d(in_ch, in_ch, kernal_size, stride=stride, padding=0),
                                  DoubleConv(in_ch, out_ch, norm_type, soft=False))

    def forward(self, x):
        return self.conv(x)


class Down(nn.Module):
    def __init__(self, in_ch, out_ch, norm_type=2, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_ch, in_ch, kernal_size, stride=stride, padding=0),
                                  DoubleConv(in_ch, out_ch, norm_type, soft=False))

    def forward(self, x):
        return self.conv(x)


def _maybe_dropout(x, p):
    return F.dropout(x, p=p, training=True) if p else x


class VAE(nn.Module):
    """Shape VAE.  ``spatial`` = input side S (multiple of 32); latent flatten = 256*(S/32)^3.

    The reference hard-wires S=128 (flatten 16384, view(B,256,4,4,4)); ``spatial=128`` gives the
    reference's state_dict shapes exactly.  ``noise`` lets a test inject the z the reference drew.
    """

    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=FMAPS, dim=1024, soft=False, spatial=128):
        super().__init__()
        f = list(n_fmaps)
        self.in_block = Conv(n_class, f[0], norm_type=norm_type)
        for i in range(5):
            setattr(self, "down%d" % (i + 1), Down(f[i], f[i + 1], norm_type=norm_type))
        self.side = spatial // 32
        self.top_ch = f[5]
        self.flat = f[5] * self.side ** 3
        self.fc_mean = nn.Linear(self.flat, dim)
        self.fc_std = nn.Linear(self.flat, dim)
        self.fc2 = nn.Linear(dim, self.flat)
        for i in range(5):
            setattr(self, "up%d" % (i + 1), Up(f[5 - i], f[4 - i], norm_type=norm_type))
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class

    def encode(self, x):
        x = self.in_block(x)
        for i in range(1, 6):
            x = getattr(self, "down%d" % i)(x)
        x = x.reshape(x.size(0), self.flat)
        return self.fc_mean(x), F.relu(self.fc_std(x))

    def decode(self, z, dropout=0.0):
        x = self.fc2(z).view(z.size(0), self.top_ch, self.side, self.side, self.side)
        for i in range(1, 6):
            x = _maybe_dropout(getattr(self, "up%d" % i)(x), dropout)
        return self.final(self.out_block(x))

    def forward(self, x, if_random=False, scale=1, mid_input=False, dropout=0.0, noise=None):
        if mid_input:
            return self.decode(x, dropout)
        mean, std = self.encode(x)
        if noise is None:
            noise = torch.randn(mean.size(0), mean.size(1))
        z = mean + noise.to(mean) * std * scale if if_random else mean
        return self.decode(z, dropout), mean, std


class Segmentation(nn.Module):
    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=FMAPS):
        super().__init__()
        f = list(n_fmaps)
        self.in_block = Conv(n_channels, f[0], norm_type=norm_type)
        for i in range(4):
            setattr(self, "down%d" % (i + 1), Down(f[i], f[i + 1], norm_type=norm_type))
        for i in range(4):
            setattr(self, "up%d" % (i + 2), Up(f[4 - i], f[3 - i], norm_type=norm_type))
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class

    def forward(self, data_dict, in_key, out_key, dropout=0.0):
        x1 = self.in_block(data_dict[in_key])
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = _maybe_dropout(self.up2(x5), dropout)
        x = _maybe_dropout(self.up3(x) + x3, dropout)
        x = _maybe_dropout(self.up4(x) + x2, dropout)
        x = _maybe_dropout(self.up5(x), dropout)
        x = _maybe_dropout(self.out_block(x), dropout)
        data_dict[out_key] = self.final(x)
        return data_dict


class Joint(nn.Module):
    def __init__(self, models, vae_forward_scale=0.0, vae_decoder_dropout=0.0, seg_dropout=0.0):
        super().__init__()
        self.Seg, self.Vae = models[0], models[1]
        self.vae_forward_scale = vae_forward_scale
        self.vae_decoder_dropout = vae_decoder_dropout
        self.seg_dropout = seg_dropout

    def forward(self, data_dict, in_key, out_key, out_key_recon, dropout=False):
        if dropout:
            data_dict = self.Seg(data_dict, in_key, out_key, dropout=self.seg_dropout)
            data_dict[out_key_recon], _, _ = self.Vae(data_dict[out_key], if_random=False,
                                                      scale=self.vae_forward_scale,
                                                      dropout=self.vae_decoder_dropout)
        else:
            data_dict = self.Seg(data_dict, in_key, out_key)
            (data_dict[out_key_recon], data_dict["mean"], data_dict["std"]) = self.Vae(
                data_dict[out_key], if_random=False, scale=self.vae_forward_scale)
        return data_dict


class Encoder(nn.Module):
    """joint_model.py:274-303 — the VAE encoder trunk + fc1/fc2/fc_mean, sigmoid output (the discriminator `Dis` of
    domain_adaptation_dis, main_target.py:338-341, and the image encoder of Embed).  ``spatial`` as in VAE (reference: 128)."""

    def __init__(self, n_channels, dim, norm_type=2, n_fmaps=FMAPS, soft=False, spatial=128):
        super().__init__()
        f = list(n_fmaps)
        self.in_block = Conv(n_channels, f[0], norm_type=norm_type)
        for i in range(5):
            setattr(self, "down%d" % (i + 1), Down(f[i], f[i + 1], norm_type=norm_type))
        self.flat = f[5] * (spatial // 32) ** 3
        self.fc1 = nn.Linear(self.flat, 1024)
        self.fc2 = nn.Linear(1024, 128)
        self.fc_mean = nn.Linear(128, dim)

    def forward(self, x):
        x = self.in_block(x)
        for i in range(1, 6):
            x = getattr(self, "down%d" % i)(x)
        x = F.relu(self.fc1(x.reshape(x.size(0), self.flat)))
        x = F.relu(self.fc2(x))
        return torch.sigmoid(self.fc_mean(x))


class Fusion(nn.Module):
    """joint_model.py:392-437 — U-Net over an image and a mask branch merged (added) at half resolution."""

    def __init__(self, n_channels_img, n_channels_mask, n_class, norm_type=2, n_fmaps=FMAPS):
        super().__init__()
        f = list(n_fmaps)
        self.in_block = Conv(n_channels_img, f[0], norm_type=norm_type)
        self.down1 = Down(f[0], f[1], norm_type=norm_type)
        self.in_block_mask = Conv(n_channels_mask, f[0], norm_type=norm_type)
        self.down1_mask = Down(f[0], f[1], norm_type=norm_type)
        self.merge = Conv(f[1], f[1], norm_type=norm_type)
        self.down2 = Down(f[1], f[2], norm_type=norm_type)
        self.down3 = Down(f[2], f[3], norm_type=norm_type)
        self.down4 = Down(f[3], f[4], norm_type=norm_type)
        self.up2 = Up(f[4], f[3], norm_type=norm_type)
        self.up3 = Up(f[3], f[2], norm_type=norm_type)
        self.up4 = Up(f[2], f[1], norm_type=norm_type)
        self.up5 = Up(f[1], f[0], norm_type=norm_type)
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class

    def forward(self, data_dict, in_key_img, in_key_mask, out_key):
        x2 = self.down1(self.in_block(data_dict[in_key_img])) + self.down1_mask(self.in_block_mask(data_dict[in_key_mask]))
        x2 = self.merge(x2)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up2(x5)
        x = self.up3(x) + x3
        x = self.up4(x) + x2
        x = self.up5(x)
        data_dict[out_key] = self.final(self.out_block(x))
        return data_dict


class Joint2(nn.Module):
    """joint_model.py:454-466 — segmenter + discriminator on the foreground probability."""

    def __init__(self, models, seg_dropout=0.0):
        super().__init__()
        self.Seg, self.Dis = models[0], models[1]
        self.seg_dropout = seg_dropout

    def forward(self, data_dict, in_key, out_key, score_key, dropout=False):
        if dropout:
            data_dict = self.Seg(data_dict, in_key, out_key, dropout=self.seg_dropout)
        else:
            data_dict = self.Seg(data_dict, in_key, out_key)
        data_dict[score_key] = self.Dis(data_dict[out_key][:, 1:2, :, :, :])
        return data_dict


class Embed(nn.Module):
    """joint_model.py:469-500 — image encoder -> latent code -> VAE decoder (initial segmentation) -> Fusion refinement."""

    def __init__(self, models):
        super().__init__()
        self.Encoder, self.Vae, self.Fusion = models[0], models[1], models[2]

    def forward(self, data_dict, in_key, out_key, test_mode=False, loop_input=None, seg_input=None, latent_input=None, noise=None):
        data_dict["latent_code"] = data_dict[latent_input] if latent_input else self.Encoder(data_dict[in_key])
        data_dict["gt_recon"], data_dict["latent_code_gt"], data_dict["latent_code_std"] = self.Vae(
            data_dict["venous_pancreas_only"], if_random=True, scale=0.5, mid_input=False, noise=noise)
        if loop_input:
            data_dict[loop_input], data_dict["latent_code_loop"], _ = self.Vae(data_dict[loop_input], if_random=False, scale=0, mid_input=False)
        if seg_input:
            data_dict["init_seg"] = data_dict[seg_input]
        else:
            data_dict["init_seg"] = self.Vae(data_dict["latent_code"], if_random=False, scale=0, mid_input=True)
        if loop_input:
            data_dict = self.Fusion(data_dict, in_key, loop_input, out_key)
        elif test_mode:
            data_dict = self.Fusion(data_dict, in_key, "init_seg", out_key)
        else:
            data_dict = self.Fusion(data_dict, in_key, "gt_recon", out_key)
        data_dict["seg_recon"], _, _ = self.Vae(data_dict["init_seg"].detach(), if_random=False, scale=0, mid_input=False)
        return data_dict


# --------------------------------------------------------------------------------------
# losses / label prep
# --------------------------------------------------------------------------------------
EPS_EVALUATION = 1e-6   # utils/evaluation.py:72-79
EPS_MAIN_SOURCE = 1e-4  # main_source.py:174-181


# --------------------------------------------------------------------------------------
# the *_GS family (joint_model.py:17-33, 54-99, 140-202, 307-346) — instantiated nowhere in the reference; restated for completeness
# --------------------------------------------------------------------------------------
class GSNorm3d(nn.Module):
    """joint_model.py:17-33: every channel divided by (the sum of its group's channels + 1e-4)"""

    def __init__(self, out_ch, num_group=1):
        super().__init__()
        self.out_ch, self.num_group = out_ch, num_group

    def forward(self, x):
        n, c = x.shape[0], x.shape[1]
        g = x.reshape(n, self.num_group, c // self.num_group, *x.shape[2:])
        return (g / (g.sum(2, keepdim=True) + 0.0001)).reshape(x.shape)


def _ca(cin, cout, soft, inplace):
    return [nn.Conv3d(cin, cout, 3, padding=1), _act(soft, inplace)]


class DoubleConv_GS(nn.Module):
    def __init__(self, in_ch, out_ch, num_group=1, soft=False):
        super().__init__()
        act = _act(soft, False)
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, 3, padding=1), act, nn.Conv3d(out_ch, out_ch, 3, padding=1), act)

    def forward(self, x):
        return self.conv(x)


class Up_GS(nn.Module):
    def __init__(self, in_ch, out_ch, num_group=1, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Upsample(scale_factor=2, mode="trilinear"), DoubleConv_GS(in_ch, out_ch, num_group, soft=False))

    def forward(self, x):
        return self.conv(x)


class Down_GS(nn.Module):
    def __init__(self, in_ch, out_ch, num_group=1, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_ch, in_ch, kernal_size, stride=stride, padding=0), DoubleConv_GS(in_ch, out_ch, num_group, soft=False))

    def forward(self, x):
        return self.conv(x)


class Conv_GS(nn.Module):
    def __init__(self, in_ch, out_ch, num_group=1, activation=True, norm=True, soft=False):
        super().__init__()
        self.conv = nn.Sequential(*_ca(in_ch, out_ch, soft, True))

    def forward(self, x):
        return self.conv(x)


def group_normalised_weight(weight, num_group):
    """joint_model.py:154-161: |w| / (its sum over each group of input channels)"""
    w = weight.abs()
    o, i = w.shape[0], w.shape[1]
    g = w.reshape(o, num_group, i // num_group, *w.shape[2:])
    return (g / g.sum(2, keepdim=True)).reshape(w.shape)


class GSConv3d(nn.Conv3d):
    def __init__(self, in_channels, out_channels, kernel_size, num_group=1, stride=1, padding=0, dilation=1, groups=1, bias=True, if_sub=None,
                 trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.weight.requires_grad = bool(trainable)
        self.num_group = num_group

    def forward(self, x):
        return F.conv3d(x, group_normalised_weight(self.weight, self.num_group), self.bias, self.stride, self.padding, self.dilation, self.groups)


class GSConvTranspose3d(nn.ConvTranspose3d):
    def __init__(self, in_channels, out_channels, kernel_size, num_group=1, stride=1, padding=0, dilation=1, output_padding=0, groups=1, bias=False,
                 if_sub=None, trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias, dilation)
        self.weight.requires_grad = bool(trainable)
        self.num_group = num_group

    def forward(self, x, output_size=None):
        return F.conv_transpose3d(x, group_normalised_weight(self.weight, self.num_group), self.bias, self.stride, self.padding,
                                  self.output_padding, self.groups, self.dilation)


class SConv3d(nn.Conv3d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, if_sub=None, trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.weight.requires_grad = bool(trainable)

    def forward(self, x):
        w = self.weight - self.weight.mean((2, 3, 4), keepdim=True)
        return F.conv3d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)


class Segmentation_GS(nn.Module):
    """joint_model.py:307-346"""

    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=FMAPS):
        super().__init__()
        f = list(n_fmaps)
        self.in_block = Conv_GS(n_channels, f[0], num_group=2)
        self.down1 = Down_GS(f[0], f[1], num_group=2)
        self.down2 = Down_GS(f[1], f[2], num_group=2)
        self.down3 = Down_GS(f[2], f[3], num_group=4)
        self.norm1, self.norm2 = GSNorm3d(f[0], 2), GSNorm3d(f[1], 4)
        self.norm3, self.norm4 = GSNorm3d(f[2], 8), GSNorm3d(f[3], 8)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear")
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear")
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear")
        self.out_block1 = Conv_GS(f[0] + f[1] + f[2] + f[3], 32)
        self.out_block2 = nn.Conv3d(32, n_class, 1, padding=0)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class

    def forward(self, data_dict, in_key, out_key):
        x1 = self.in_block(data_dict[in_key])
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        levels = (self.norm1(x1), self.up2(self.norm2(x2)), self.up4(self.norm3(x3)), self.up8(self.norm4(x4)))
        data_dict[out_key] = self.final(self.out_block2(self.out_block1(torch.cat(levels, dim=1))))
        return data_dict


def one_hot(label, n_class=2):
    """(B,1,D,H,W) integer-valued labels -> (B,n_class,D,H,W) float one-hot."""
    lab = label.long()
    out = torch.zeros(lab.size(0), n_class, *lab.shape[2:], dtype=torch.float32)
    return out.scatter_(1, lab, 1)


def _hard(mask):
    idx = torch.argmax(mask, dim=1, keepdim=True)
    return torch.zeros_like(mask).scatter_(1, idx, 1)


def dice_scores(s, t, eps=EPS_EVALUATION):
    """per (b,c) soft dice 2*sum(s*t)/(sum s + sum t + eps); -> (B,C)."""
    dims = (2, 3, 4)
    return 2 * torch.sum(s * t, dims) / (torch.sum(s, dims) + torch.sum(t, dims) + eps)


def avg_dsc(data_dict, source_key="align_lung", target_key="source_lung", binary=False, topindex=2,
            botindex=0, pad=(0, 0, 0), return_mean=True, detach=False, eps=EPS_EVALUATION):
    s, t = data_dict[source_key], data_dict[target_key]
    if detach:
        t = t.detach()
    if binary:
        s, t = _hard(s), _hard(t)
    d = dice_scores(s, t, eps)
    if s.shape[1] > 1:
        d = d[:, botindex:topindex]
        return torch.mean(d) if return_mean else torch.mean(d, 1)
    return torch.mean(d) if return_mean else torch.mean(d, 1)


def dice(a, b):
    return 2.0 * torch.sum(a * b) / (torch.sum(a) + torch.sum(b) + 1e-6)


def KLloss(data_dict, mean_key="mean", std_key="std"):
    m, s = data_dict[mean_key], data_dict[std_key]
    per_sample = 0.5 * ((s * s).sum(1) + (m * m).sum(1) - 2 * torch.log(s + 1e-5).sum(1))
    return per_sample.mean()


def binarize(a):
    return (a >= 0.5).float()


def confident_binarize(a, max=0.8, min=0.2):
    b = a.clone()
    b[b > max] = 1
    b[b < min] = 0
    return b


def avg_ce(data_dict, source_key="align_lung", target_key="source_lung"):
    src = data_dict[source_key]
    if not isinstance(src, list):
        src = [src]
    crit = nn.BCELoss()
    return sum(crit(im, data_dict[target_key]) for im in src) / len(src)


# --------------------------------------------------------------------------------------
# train-step loss bodies (what the benchmark times, minus data loading / logging)
# --------------------------------------------------------------------------------------
def joint_train_losses(joint, img, label, lambda_vae=0.1, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img, "gt": one_hot(label, n_class)}
    batch = joint(batch, "img", "pred", "recon")
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    final = lambda_vae * recon_loss + dsc_loss
    return final, {"recon_loss": recon_loss, "dice_loss": dsc_loss, "batch": batch}


def seg_train_losses(seg, img, label, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img, "gt": one_hot(label, n_class)}
    batch = seg(batch, "img", "pred")
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    return dsc_loss, {"dice_loss": dsc_loss, "batch": batch}


def vae_train_losses(vae, label, scale=0.35, noise=None, eps=EPS_MAIN_SOURCE, n_class=2):
    gt = one_hot(label, n_class)
    recon, mean, std = vae(gt, if_random=True, scale=scale, noise=noise)
    batch = {"gt": gt, "recon": recon, "mean": mean, "std": std}
    kl = KLloss(batch)
    dsc_loss = 1 - avg_dsc(batch, "recon", "gt", botindex=1, topindex=n_class, eps=eps)
    final = dsc_loss + 0.00002 * kl
    return final, {"dice_loss": dsc_loss, "kl_loss": kl, "batch": batch}


def lambda_schedule(recon_loss, lambda_vae):
    """main_target.py:551-554 (domain_loss_type 8/9/15/16)."""
    r = float(recon_loss)
    if r < 0.15:
        return lambda_vae * 0.6
    if r < 0.225:
        return lambda_vae * 1.2
    if r < 0.3:
        return lambda_vae * 2.0
    return lambda_vae * 3.0


def domain_adaptation_losses(student, teacher, img, label, lambda_vae=1.0, domain_loss_type=0, kl=False,
                             use_confident_binarize=False, eps=EPS_EVALUATION, n_class=2):
    """One Monte-Carlo pass (vae_mont_number=1) of main_target.py:531-596."""
    batch = {"img": img, "gt": one_hot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    with torch.no_grad():
        batch = teacher(batch, "img", "fake", "_unused")
    fake = batch["fake"]
    batch["fake"] = confident_binarize(fake) if use_confident_binarize else binarize(fake)
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    klloss = KLloss(batch)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    if domain_loss_type == 8:
        cur = lambda_schedule(recon_loss, lambda_vae)
        if cur > 1:
            final = recon_loss + (klloss if kl else 0) + 1 / cur * fake_loss
        else:
            final = cur * (recon_loss + (klloss if kl else 0)) + fake_loss
    elif domain_loss_type == 9:
        cur = lambda_schedule(recon_loss, lambda_vae)
        final = (cur * recon_loss + fake_loss) / (1 + cur)
    elif domain_loss_type == 0:
        final = lambda_vae * recon_loss + fake_loss
        if kl:
            final = final + 0.00002 * lambda_vae * klloss
    else:
        raise ValueError("oracle restates domain_loss_type 0, 8, 9")
    return final, {"recon_loss": recon_loss, "kl_loss": klloss, "dice_loss": dsc_loss,
                   "dice_loss_fake": fake_loss, "batch": batch}


def finetune_loss(recon_loss, fake_loss, klloss, lambda_vae=1.0, domain_loss_type=0, kl=False, only_pseudo=False):
    """Loss of one test-time-training iteration, main_target.py:835-884 — the branches the shipped scripts reach:
    only_pseudo (:835-836), domain_loss_type 8 (:837-847) and 9 (:848-853), and the default
    `lambda_vae * recon_loss + dsc_loss_fake` (:881-882; epoch >= lambda_vae_warmup, turn_epoch == -1)."""
    if only_pseudo:
        return fake_loss
    if domain_loss_type == 8:
        cur = lambda_schedule(recon_loss, lambda_vae)
        if cur > 1:
            return recon_loss + (klloss if kl else 0) + 1 / cur * fake_loss
        return cur * (recon_loss + (klloss if kl else 0)) + fake_loss
    if domain_loss_type == 9:
        cur = lambda_schedule(recon_loss, lambda_vae)
        return (cur * recon_loss + fake_loss) / (1 + cur)
    if domain_loss_type == 0:
        return lambda_vae * recon_loss + fake_loss
    raise ValueError("oracle restates the finetune loss for only_pseudo and domain_loss_type 0, 8, 9")


def embed_train_losses(embed, img, label, eps=EPS_MAIN_SOURCE, n_class=2, noise=None):
    """main_source.py:546-590."""
    gt = one_hot(label, n_class).to(img.dtype)
    batch = embed({"img": img, "venous_pancreas_only": gt}, "img", "pred", test_mode=True, noise=noise)
    batch["gt"] = gt
    d = lambda key: 1 - avg_dsc(batch, key, "gt", botindex=1, topindex=n_class, eps=eps)
    dsc1, dsc2, recon, inpaint = d("pred"), d("init_seg"), d("gt_recon"), d("seg_recon")
    kl = KLloss(batch, mean_key="latent_code_gt", std_key="latent_code_std")
    mse = torch.nn.functional.mse_loss(batch["latent_code"], batch["latent_code_gt"])
    final = (dsc1 + dsc2 + inpaint) / 3 + mse / 10 + 0.00002 * kl + recon
    return final, {"dice_loss1": dsc1, "dice_loss2": dsc2, "mse_loss": mse, "kl_loss": kl, "recon_loss": recon, "inpaint_loss": inpaint, "batch": batch}


def refine_vae_losses(embed, img, label, eps=EPS_MAIN_SOURCE, n_class=2, noise=None):
    """main_source.py:591-628."""
    gt = one_hot(label, n_class).to(img.dtype)
    batch = embed({"img": img, "venous_pancreas_only": gt}, "img", "pred", test_mode=True, noise=noise)
    batch["gt"] = gt
    d = lambda key: 1 - avg_dsc(batch, key, "gt", botindex=1, topindex=n_class, eps=eps)
    recon, inpaint = d("gt_recon"), d("seg_recon")
    kl = KLloss(batch, mean_key="latent_code_gt", std_key="latent_code_std")
    final = inpaint + 0.00002 * kl + recon
    return final, {"recon_loss": recon, "inpaint_loss": inpaint, "kl_loss": kl, "batch": batch}


def sep_joint_train_losses(joint, teacher, img, label, eps=EPS_MAIN_SOURCE, n_class=2):
    """main_source.py:629-659."""
    batch = {"img": img, "gt": one_hot(label, n_class).to(img.dtype)}
    batch = joint(batch, "img", "pred", "recon")
    with torch.no_grad():
        tb = teacher({"img": img}, "img", "pred_tea", "recon_tea")
    batch["pred_tea"], batch["recon_tea"] = tb["pred_tea"], tb["recon_tea"]
    kw = dict(botindex=1, topindex=n_class, return_mean=False, eps=eps)
    recon = avg_dsc(batch, "pred", "recon", **kw)
    recon_tea = avg_dsc(batch, "pred_tea", "recon_tea", **kw)
    dsc = avg_dsc(batch, "pred", "pred_tea", **kw)
    final = 0.1 * (1 - torch.mean(recon)) + 1 - torch.mean(dsc * recon_tea ** 2)
    return final, {"recon_loss": 1 - torch.mean(recon), "dice_loss": 1 - torch.mean(dsc), "batch": batch}


def discriminator_train_loss(dis, mask, score):
    """main_target.py:491-501."""
    out = dis(mask)
    final = torch.square(score.to(out) - out).mean()
    return final, {"final_loss": final, "score_out": out}


def domain_adaptation_dis_losses(student, teacher_seg, img, label, lambda_vae=1.0, epoch=1, lambda_vae_warmup=0, eps=EPS_EVALUATION, n_class=2):
    """main_target.py:696-732."""
    batch = {"img": img, "gt": one_hot(label, n_class).to(img.dtype)}
    batch = student(batch, "img", "pred", "score", dropout=True)
    with torch.no_grad():
        batch = teacher_seg(batch, "img", "fake")
    batch["fake"] = binarize(batch["fake"])
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    dis_loss = 1 - batch["score"].mean()
    lam = lambda_vae if epoch >= lambda_vae_warmup else lambda_vae * epoch / lambda_vae_warmup
    final = lam * dis_loss + fake_loss
    return final, {"discriminator_loss": dis_loss, "dice_loss_fake": fake_loss, "dice_loss": dsc_loss, "batch": batch}


def test_time_finetune(model, model_ft, teacher, img, label, steps, lr=1e-2, weight_decay=0.0, lambda_vae=1.0,
                       domain_loss_type=0, kl=False, only_pseudo=False, use_confident_binarize=False, n_class=2):
    """Per-case test-time training, main_target.py:809-900: model_ft starts from model's weights (:811), then `steps`
    iterations of [student = model_ft forward (dropout flag on, rates as configured), frozen teacher forward, binarised pseudo-label,
    three Dice terms (evaluation eps), loss above, a fresh SGD(lr, weight_decay, momentum=0) step (:886-891)].
    -> list of per-iteration dicts of loss scalars (as the reference logs them, :893-897)."""
    model_ft.load_state_dict(model.state_dict())
    log = []
    for _ in range(steps):
        batch = {"img": img, "gt": one_hot(label, n_class).to(img.dtype)}
        batch = model_ft(batch, "img", "pred", "recon", dropout=True)
        batch = teacher(batch, "img", "fake", "_unused")            # not under no_grad in the reference; the teacher is frozen
        klloss = KLloss(batch)
        fake = batch["fake"]
        batch["fake"] = confident_binarize(fake) if use_confident_binarize else binarize(fake)
        recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
        dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
        fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
        final = finetune_loss(recon_loss, fake_loss, klloss, lambda_vae, domain_loss_type, kl, only_pseudo)
        opt = torch.optim.SGD([p for p in model_ft.parameters() if p.requires_grad], lr=lr, weight_decay=weight_decay, momentum=0)
        opt.zero_grad()
        final.backward()
        opt.step()
        log.append({"recon_loss": recon_loss.detach(), "dice_loss_fake": fake_loss.detach(), "dice_loss": dsc_loss.detach(),
                    "final_loss": final.detach()})
    return log


# --------------------------------------------------------------------------------------
# deterministic, RNG-free parameter fill and synthetic inputs (shared by goldens, tests, bench)
# --------------------------------------------------------------------------------------
def _mix64(x):
    """splitmix64 finaliser on uint64 numpy arrays (wraps mod 2^64)."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hashed_uniform(n, stream, seed=0):
    """n floats in [0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) % (1 << 64))
        h = _mix64(idx * np.uint64(0x9E3779B97F4A7C15) + key)
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def kat_scores(seed, n_class=4, shape=(2, 3, 4, 5)):
    """(B, n_class, D, H, W) hashed scores for the multi-class hard-Dice known-answer test (tests/golden/kats.npz: dice4_*), with two planted
    ties — every channel equal at one voxel (argmax -> channel 0) and channels 1 and 2 sharing the maximum at another (-> channel 1)."""
    b = shape[0]
    x = torch.from_numpy(hashed_uniform(b * n_class * int(np.prod(shape[1:])), 8101, seed)).view(b, n_class, *shape[1:]).clone()
    x[0, :, 0, 0, 0] = 0.25
    x[b - 1, :, 1, 1, 1] = 0.1
    x[b - 1, 1:3, 1, 1, 1] = 0.9
    return x


def _name_stream(name):
    return zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF


def _fan_in(name, p):
    if p.dim() >= 2:
        return int(np.prod(p.shape[1:]))
    return max(int(p.numel()), 1)


def deterministic_fill_(module, seed=0, gain=1.0):
    """Fill every parameter with uniform(-a, a), a = gain*sqrt(3/fan_in) for weights and a = 0.1 for
    biases.  No RNG: values depend only on (seed, crc32(parameter name), element index), so the same call
    on the reference modules, on this oracle and on the HIP modules gives identical weights."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            u = hashed_uniform(p.numel(), _name_stream(name), seed)
            bound = 0.1 if p.dim() == 1 else gain * math.sqrt(3.0 / _fan_in(name, p))
            vals = (2.0 * u - 1.0) * np.float32(bound)
            p.copy_(torch.from_numpy(vals).view_as(p))
    return module


def positive_fill_(module):
    """After deterministic_fill_: every parameter replaced by its absolute value.  For the GSNorm3d blocks (norm_type 3): the layer divides by the
    per-voxel channel SUM + 1e-4, which is ill-conditioned wherever mixed-sign channels cancel; positive weights on positive inputs keep the
    sums away from zero so that a fixture pins arithmetic, not the position of near-singular voxels.  Same call on the reference modules and the native ones."""
    with torch.no_grad():
        for p in module.parameters():
            p.abs_()
    return module


def bn_fill_(module):
    """After deterministic_fill_: BatchNorm3d scales become 1 + 3 * (their +-0.1 fill) = 0.7 .. 1.3 (a scale around 0 would test nothing);
    shifts keep their +-0.1 fill, running statistics their defaults (0, 1).  Same call on the reference modules and the native ones."""
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.weight.mul_(3.0).add_(1.0)
    return module


def synthetic_image(batch, side, seed=2):
    """~ clip(N(0,1), -1, 1), shape (B,1,S,S,S); Box-Muller on the hashed uniforms."""
    n = batch * side ** 3
    u1 = np.maximum(hashed_uniform(n, 1001, seed), 1e-7).astype(np.float64)
    u2 = hashed_uniform(n, 1002, seed).astype(np.float64)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(np.clip(g, -1, 1).astype(np.float32)).view(batch, 1, side, side, side)


def synthetic_label(batch, side, seed=3, kind="ellipsoid", n_class=2):
    """(B,1,S,S,S) float labels in {0..n_class-1}: a centred ellipsoid with a hashed ragged rim (n_class > 2: nested ellipsoidal shells,
    label k inside the k-th), or Bernoulli(0.1)."""
    n = batch * side ** 3
    u = hashed_uniform(n, 2001, seed).reshape(batch, side, side, side)
    if kind == "bernoulli":
        lab = (u < 0.1).astype(np.float32)
    else:
        ax = (np.arange(side, dtype=np.float32) + 0.5) / side - 0.5
        z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
        r = (z / 0.30) ** 2 + (y / 0.22) ** 2 + (x / 0.36) ** 2
        rr = r[None] + 0.35 * (u - 0.5)
        lab = np.zeros(rr.shape, dtype=np.float32)
        for k in range(1, n_class):
            lab += rr < ((n_class - k) / (n_class - 1.0)) ** 2          # k = 1: the threshold 1.0 of the two-class volume
    return torch.from_numpy(lab.astype(np.float32)).view(batch, 1, side, side, side)


def build_joint(spatial, dim=128, seed=0, n_class=2):
    seg = Segmentation(n_channels=1, n_class=n_class, norm_type=1)
    vae = VAE(n_channels=n_class, n_class=n_class, norm_type=1, dim=dim, spatial=spatial)
    joint = Joint([seg, vae])
    deterministic_fill_(joint, seed=seed)
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    return joint
