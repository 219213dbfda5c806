"""The nn.Module surface of the reference's joint_model.py, executed by the libvaeseg HIP kernels.

Class names, constructor / forward signatures, attribute names (``.Seg`` / ``.Vae``, ``in_block``, ``down1`` ...) and
state_dict keys / shapes follow /root/reference/joint_model.py (cited per class) so checkpoints and the
main_source.py / main_target.py loops keep working.  Parameters live in ordinary ``nn.Conv3d`` /
``nn.ConvTranspose3d`` / ``nn.Linear`` holders placed at the same Sequential indices as in the reference
(InstanceNorm3d / ReLU entries are parameter-less there too), but ``forward`` never calls those holders:
it launches the fused kernels through ``ops``.  There is no CPU path — inputs must be CUDA tensors.

The configuration every entry point of the reference uses — ``norm_type=1`` (InstanceNorm3d), ReLU (``soft=False``), two classes
 — runs fused: normalisation and activation are applied inside the consuming conv kernels.  The constructors' other
settings, ``norm_type=2`` (BatchNorm3d, the class default, with affine parameters and running statistics) and ``soft=True``
(Softplus), run through ``ops.NormAct``: native too, but as separate streaming passes after each conv (nothing in the reference
reaches them: main_source.py:250-272, main_target.py:317-342 pass norm_type=1, and every block receives soft=False).
The ``*_GS`` classes, which no reference code instantiates, live in ``modules_gs.py``; ``norm_type=3`` (GSNorm3d with one group) inside these blocks runs as three native passes.
"""
import torch
import torch.nn as nn

from . import ops

_DEFAULT_DTYPE = torch.float32


def set_default_kernel_dtype(dtype):
    """fp32 (parity mode, exact-f32 MFMA), bf16 or fp16 (throughput modes, fp32 accumulate; fp16 needs optim.LossScaler)."""
    global _DEFAULT_DTYPE
    if dtype not in ops.KERNEL_DTYPES:
        raise TypeError("kernel dtype must be torch.float32, torch.bfloat16 or torch.float16")
    _DEFAULT_DTYPE = dtype


def set_kernel_dtype(module, dtype):
    if dtype not in ops.KERNEL_DTYPES:
        raise TypeError("kernel dtype must be torch.float32, torch.bfloat16 or torch.float16")
    for m in module.modules():
        if hasattr(m, "kernel_dtype"):
            m.kernel_dtype = dtype
    return module


class Act:
    """A channels-last activation travelling between blocks: ``raw`` (N,D,H,W,C) and, when the
    InstanceNorm+ReLU that follows its producer has not been applied yet, the producer's ``stats``."""
    __slots__ = ("raw", "stats")

    def __init__(self, raw, stats=None):
        self.raw, self.stats = raw, stats


def Normalization(norm_type, out_channels, num_group=1):
    """joint_model.py:9-15 — the holder at the reference's Sequential index: parameter-less InstanceNorm3d, or BatchNorm3d whose
    weight / bias / running_mean / running_var / num_batches_tracked are the state the native kernels read and update."""
    if norm_type == 1:
        return nn.InstanceNorm3d(out_channels)
    if norm_type == 2:
        return nn.BatchNorm3d(out_channels, momentum=0.1)
    if norm_type == 3:
        return GSNormHolder(out_channels, num_group)
    raise ValueError("norm_type must be 1 (InstanceNorm3d), 2 (BatchNorm3d) or 3 (GSNorm3d)")


class GSNormHolder(nn.Module):
    """joint_model.py:17-33 at a block's Sequential index (norm_type=3): parameter-less; x[:, group] / (sum over the group's channels + 1e-4).
    The reference's Conv / DoubleConv never forward num_group (joint_model.py:41,44,47,107), so inside these blocks it is always one group.
    Native through vs_gsnorm_* (ops.GSNorm), as the `*_GS` family's own GSNorm3d (modules_gs.py)."""

    def __init__(self, out_ch, num_group=1):
        super().__init__()
        self.out_ch, self.num_group = out_ch, num_group


def _activation(soft, inplace):
    """joint_model.py:38,104 — the holder at the reference's Sequential index"""
    return nn.Softplus() if soft else nn.ReLU(inplace=inplace)


def _as_act(x, dtype):
    """Accept a planar NCDHW tensor at a block boundary (tests / ad-hoc use) or an Act."""
    if isinstance(x, Act):
        return x, False
    ops._require_cuda(x)
    return Act(ops.PackPlanar.apply(x, dtype), None), True


def _as_tensor(a, channels):
    return ops.UnpackPlanar.apply(ops.Materialize.apply(a.raw, a.stats, None, None), channels)


def _conv3(conv, a):
    y, ys = ops.ConvK3.apply(a.raw, a.stats, conv.weight, conv.bias)
    return Act(y, ys)


def _conv_norm_act(seq, i, a):
    """Sequential entries i, i+1, i+2 = conv3x3x3, normalisation, activation (joint_model.py:40-48,106-108).  InstanceNorm + ReLU stays
    lazy (applied by the consuming kernels); BatchNorm and / or Softplus go through ops.NormAct and leave as a stored activation."""
    conv, norm, actm = seq[i], seq[i + 1], seq[i + 2]
    if isinstance(norm, nn.InstanceNorm3d) and isinstance(actm, nn.ReLU):
        return _conv3(conv, a)
    y, ys = ops.ConvK3.apply(a.raw, a.stats, conv.weight, conv.bias, True)
    if isinstance(norm, GSNormHolder):
        # conv (live bias: nothing cancels it) -> group-sum normalisation -> activation, three native passes
        if y.shape[-1] != norm.out_ch:
            raise ValueError("GSNorm3d(%d) got %d stored channels" % (norm.out_ch, y.shape[-1]))
        z = ops.GSNorm.apply(y, norm.num_group)
        act = ops.VS_ACT_SOFTPLUS if isinstance(actm, nn.Softplus) else ops.VS_ACT_RELU
        return Act(ops.NormAct.apply(z, ys, None, None, "none", act, conv.weight.shape[0]), None)
    bn = norm if isinstance(norm, nn.BatchNorm3d) else None
    act = ops.VS_ACT_SOFTPLUS if isinstance(actm, nn.Softplus) else ops.VS_ACT_RELU
    out = ops.NormAct.apply(y, ys, bn.weight if bn is not None else None, bn.bias if bn is not None else None, bn, act, conv.weight.shape[0])
    return Act(out, None)


class DoubleConv(nn.Module):
    """joint_model.py:35-52 — three (conv3x3x3 -> norm -> act) triples, Sequential indices 0..8."""

    def __init__(self, in_ch, out_ch, norm_type=2, soft=False):
        super().__init__()
        activation = _activation(soft, False)               # one module at indices 2, 5, 8, as in the reference
        layers = []
        for cin in (in_ch, out_ch, out_ch):
            layers += [nn.Conv3d(cin, out_ch, 3, padding=1), Normalization(norm_type, out_ch), activation]
        self.conv = nn.Sequential(*layers)
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x, _start=0):
        """_start = 3: x is already the (lazy) output of the first conv triple — Up ran it composed with its transposed conv (ops.UpConvK3)"""
        a, wrapped = _as_act(x, self.kernel_dtype)
        idx = [i for i in (0, 3, 6) if i >= _start]
        if self.lazy_all() and ops.chain_ok(a.raw, a.stats, [self.conv[i] for i in idx]):
            # the small volumes of the deep levels: the block's convolutions as ONE launch each way (ops.ConvK3Chain, csrc/chain.h)
            params = []
            for i in idx:
                params += [self.conv[i].weight, self.conv[i].bias]
            y, ys = ops.ConvK3Chain.apply(a.raw, a.stats, *params)
            a = Act(y, ys)
            return _as_tensor(a, self.out_ch) if wrapped else a
        for i in idx:
            a = _conv_norm_act(self.conv, i, a)
            if i < 6 and a.stats is not None:
                ops.mark_defer_apply(a.raw, self.conv[i])      # consumed once, by the next 3x3x3 conv: its IN-backward apply can be fused (ops._LAZY_APPLY)
        return _as_tensor(a, self.out_ch) if wrapped else a

    def lazy_all(self):
        """InstanceNorm + ReLU after every conv (the configuration every entry point of the reference uses)?"""
        return all(isinstance(self.conv[i + 1], nn.InstanceNorm3d) and isinstance(self.conv[i + 2], nn.ReLU) for i in (0, 3, 6))

    def lazy_head(self):
        """InstanceNorm + ReLU after the first conv (the fused configuration)?"""
        return isinstance(self.conv[1], nn.InstanceNorm3d) and isinstance(self.conv[2], nn.ReLU)


class Conv(nn.Module):
    """joint_model.py:101-112 — conv3x3x3 -> norm -> ReLU."""

    def __init__(self, in_ch, out_ch, norm_type=2, num_group=1, activation=True, norm=True, soft=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, 3, padding=1), Normalization(norm_type, out_ch),
                                  _activation(soft, True))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        a = _conv_norm_act(self.conv, 0, a)
        return _as_tensor(a, self.out_ch) if wrapped else a


class Up(nn.Module):
    """joint_model.py:114-124 — ConvTranspose3d(in,in,2,2) -> DoubleConv(in,out)."""

    def __init__(self, in_ch, out_ch, norm_type=2, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        if tuple(kernal_size) != (2, 2, 2) or tuple(stride) != (2, 2, 2):
            raise NotImplementedError("native transposed conv is written for kernel 2, stride 2")
        self.conv = nn.Sequential(nn.ConvTranspose3d(in_ch, in_ch, kernal_size, stride=stride, padding=0),
                                  DoubleConv(in_ch, out_ch, norm_type, soft=False))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        t, dc = self.conv[0], self.conv[1]
        if dc.lazy_head() and ops.up_composed_ok(a.raw, t, dc.conv[0]):
            # transposed conv + first 3x3x3 conv as one operator on the coarse grid: no intermediate tensor, one launch each way
            y, ys = ops.UpConvK3.apply(a.raw, a.stats, t.weight, t.bias, dc.conv[0].weight, dc.conv[0].bias)
            a = dc(Act(y, ys), _start=3)
        else:
            a = dc(Act(ops.ConvT2S2.apply(a.raw, a.stats, t.weight, t.bias), None))
        return _as_tensor(a, self.out_ch) if wrapped else a


class Down(nn.Module):
    """joint_model.py:126-136 — Conv3d(in,in,2,stride 2) -> DoubleConv(in,out)."""

    def __init__(self, in_ch, out_ch, norm_type=2, kernal_size=(2, 2, 2), stride=(2, 2, 2), soft=False):
        super().__init__()
        if tuple(kernal_size) != (2, 2, 2) or tuple(stride) != (2, 2, 2):
            raise NotImplementedError("native strided conv is written for kernel 2, stride 2")
        self.conv = nn.Sequential(nn.Conv3d(in_ch, in_ch, kernal_size, stride=stride, padding=0),
                                  DoubleConv(in_ch, out_ch, norm_type, soft=False))
        self.out_ch = out_ch
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        a, wrapped = _as_act(x, self.kernel_dtype)
        c = self.conv[0]
        a = self.conv[1](Act(ops.ConvK2S2.apply(a.raw, a.stats, c.weight, c.bias), None))
        return _as_tensor(a, self.out_ch) if wrapped else a


_RECOMPUTE = [False]
_RECOMPUTE_SAVED_GROUPING = [True]


def set_recompute(enabled=True):
    """Activation recomputation ("checkpointing", BASELINE configs[4]) for the Down / Up blocks of Segmentation and VAE: a block keeps only
    its input (raw tensor + statistics) through the forward pass and re-runs its kernels when backward reaches it.  Off by default — 160^3,
    batch 2, fp16 peaks at 3.5 GB of the 288 GB, and the step is bound by the bandwidth recomputation spends — but the switch exists
    (entry points: --recompute) for volumes that need the memory.  In the deterministic build the recomputed statistics are bit-identical
    to the first pass, so are the gradients (tests/test_gpu_fp16.py).  Costs launches as well as bandwidth: the weight gradients leave the
    grouped end-of-pass launches (they would pin every operand until the end)."""
    enabled = bool(enabled)
    if enabled == _RECOMPUTE[0]:
        return                          # nothing to switch: in particular set_recompute(False) on a fresh process leaves VS_WGRAD_GROUP / set_wgrad_grouping alone
    _RECOMPUTE[0] = enabled
    # the grouped weight gradients are deferred to the end of backward and keep every layer's operands alive until then — exactly the memory
    # recomputation is meant to free: in this mode each layer's weight gradient is launched where its backward runs; switching recomputation off
    # restores whatever grouping was in force before it was switched on
    if enabled:
        _RECOMPUTE_SAVED_GROUPING[0] = bool(ops._GROUP["enabled"])
        ops.set_wgrad_grouping(False)
    else:
        ops.set_wgrad_grouping(_RECOMPUTE_SAVED_GROUPING[0])


def _has_training_batchnorm(blk):
    return any(isinstance(m, nn.BatchNorm3d) and m.training for m in blk.modules())


def _run_block(blk, a):
    """blk(a) for a Down / Up block on a lazy activation; under set_recompute the block's interior activations are not kept."""
    if not (_RECOMPUTE[0] and torch.is_grad_enabled() and a.raw.requires_grad):
        return blk(a)
    if _has_training_batchnorm(blk):
        # BatchNorm3d in training mode updates running_mean / running_var / num_batches_tracked in its forward (ops.NormAct): the recomputation
        # pass would apply the momentum update a second time per step.  Such blocks keep their activations (no reference entry point builds them:
        # every script passes norm_type=1).
        return blk(a)
    from torch.utils.checkpoint import checkpoint

    def run(raw, stats):
        out = blk(Act(raw, stats))
        return out.raw, out.stats

    marks = getattr(a.raw, "_vs_defer_apply", False)

    def run_marked(raw, stats):
        if marks:
            raw._vs_defer_apply = True
        return run(raw, stats)

    raw, stats = checkpoint(run_marked, a.raw, a.stats, use_reentrant=False, preserve_rng_state=False)
    return Act(raw, stats)


def _check_n_class(n_class):
    """n_class = 1 + the number of labelled structures (main_source.py:92-93).  The probabilities, their gradient and the VAE's input travel in one
    8-channel fragment, so up to 8 classes are native (two classes — every BASELINE configuration — through the fused out_block epilogue)."""
    if not 1 <= int(n_class) <= 8:
        raise NotImplementedError("native softmax / label kernels hold n_class in one 8-channel fragment: 1..8, got %r" % (n_class,))


def _dropout(a, p):
    """F.dropout(x, p, training=True) after an Up block (joint_model.py:256-264,379-385): the lazy activation is
    materialised, masked and scaled; p == 0 (the reference default, main_target.py:70-71) costs nothing."""
    if not p:
        return a
    x = ops.Materialize.apply(a.raw, a.stats, None, None) if a.stats is not None else a.raw
    return Act(ops.Dropout.apply(x, float(p), ops.next_dropout_seed()), None)


class VAE(nn.Module):
    """joint_model.py:204-272.  ``spatial`` generalises the reference's hard-wired 128^3 input
    (Linear(16384, dim), view(B,256,4,4,4)); the default reproduces the reference state_dict exactly.
    ``noise`` (optional, (B, dim)) replaces the reference's CPU torch.randn draw (joint_model.py:246)."""

    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=[8, 16, 32, 64, 128, 256], dim=1024, soft=False,
                 spatial=128):
        super().__init__()
        _check_n_class(n_class)
        if spatial % 32 or spatial < 64:
            raise ValueError("spatial must be a multiple of 32 and >= 64 (InstanceNorm needs > 1 voxel at down5)")
        f = list(n_fmaps)
        self.in_block = Conv(n_class, f[0], norm_type=norm_type, soft=False)
        self.down1 = Down(f[0], f[1], norm_type=norm_type, soft=False)
        self.down2 = Down(f[1], f[2], norm_type=norm_type, soft=False)
        self.down3 = Down(f[2], f[3], norm_type=norm_type, soft=False)
        self.down4 = Down(f[3], f[4], norm_type=norm_type, soft=False)
        self.down5 = Down(f[4], f[5], norm_type=norm_type, soft=False)
        self.side = spatial // 32
        self.top_ch = f[5]
        flat = f[5] * self.side ** 3
        self.fc_mean = nn.Linear(flat, dim)
        self.fc_std = nn.Linear(flat, dim)
        self.fc2 = nn.Linear(dim, flat)
        self.up1 = Up(f[5], f[4], norm_type=norm_type, soft=False)
        self.up2 = Up(f[4], f[3], norm_type=norm_type, soft=False)
        self.up3 = Up(f[3], f[2], norm_type=norm_type, soft=False)
        self.up4 = Up(f[2], f[1], norm_type=norm_type, soft=False)
        self.up5 = Up(f[1], f[0], norm_type=norm_type, soft=False)
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class
        self.spatial = spatial
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x, if_random=False, scale=1, mid_input=False, dropout=0.0, noise=None):
        ops._require_cuda(x)
        if not mid_input:
            if x.shape[-1] != self.spatial:
                raise ValueError("VAE built for spatial=%d got input side %d" % (self.spatial, x.shape[-1]))
            ops.stats_arena_begin(x.device)
            a = Act(ops.planar_input(x, self.kernel_dtype), None)     # Segmentation's prediction arrives with its channels-last copy
            a = self.in_block(a)
            if a.stats is not None:
                ops.mark_defer_apply(a.raw, self.in_block.conv[0])              # in_block's output feeds down1's strided conv only
            for blk in (self.down1, self.down2, self.down3, self.down4, self.down5):
                a = _run_block(blk, a)
                if blk is not self.down5 and a.stats is not None:
                    ops.mark_defer_apply(a.raw, blk.conv[1].conv[6])            # a Down block's last conv feeds the next block's strided conv only (no skips in the VAE)
            feat = ops.Materialize.apply(a.raw, a.stats, None, None)
            x_mean, x_std = ops.LinearCLPair.apply(feat, self.fc_mean.weight, self.fc_mean.bias, False,
                                                   self.fc_std.weight, self.fc_std.bias, True)
            if if_random:
                if noise is None:
                    noise = torch.randn(x_mean.shape, device=x_mean.device, dtype=torch.float32)
                z = ops.Reparam.apply(x_mean, x_std, noise.to(x_mean.device, torch.float32).contiguous(), scale)
            else:
                z = x_mean
        else:
            ops.stats_arena_begin(x.device)
            z = x
        h = ops.LinearToCL.apply(z, self.fc2.weight, self.fc2.bias, self.top_ch, self.side, self.kernel_dtype)
        a = Act(h, None)
        for blk in (self.up1, self.up2, self.up3, self.up4, self.up5):
            a = _dropout(_run_block(blk, a), dropout)
            if a.stats is not None:
                ops.mark_defer_apply(a.raw, blk.conv[1].conv[6])             # an Up block's last conv feeds the next block's transposed conv / out_block only
        recon = ops.ConvK3Softmax.apply(a.raw, a.stats, self.out_block.weight, self.out_block.bias)
        if not mid_input:
            return recon, x_mean, x_std
        return recon


class Segmentation(nn.Module):
    """joint_model.py:349-390 — U-Net with additive skips at up3 / up4, dict-in / dict-out."""

    def __init__(self, n_channels, n_class, norm_type=2, n_fmaps=[8, 16, 32, 64, 128, 256]):
        super().__init__()
        _check_n_class(n_class)
        f = list(n_fmaps)
        self.in_block = Conv(n_channels, f[0], norm_type=norm_type, soft=False)
        self.down1 = Down(f[0], f[1], norm_type=norm_type, soft=False)
        self.down2 = Down(f[1], f[2], norm_type=norm_type, soft=False)
        self.down3 = Down(f[2], f[3], norm_type=norm_type, soft=False)
        self.down4 = Down(f[3], f[4], norm_type=norm_type, soft=False)
        self.up2 = Up(f[4], f[3], norm_type=norm_type, soft=False)
        self.up3 = Up(f[3], f[2], norm_type=norm_type, soft=False)
        self.up4 = Up(f[2], f[1], norm_type=norm_type, soft=False)
        self.up5 = Up(f[1], f[0], norm_type=norm_type, soft=False)
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, data_dict, in_key, out_key, dropout=0.0):
        x = data_dict[in_key]
        ops._require_cuda(x)
        if any(s % 16 for s in x.shape[2:]):
            raise ValueError("Segmentation needs spatial sizes that are multiples of 16, got %s" % (tuple(x.shape[2:]),))
        ops.stats_arena_begin(x.device)
        a = Act(ops.PackPlanar.apply(x, self.kernel_dtype), None)
        x1 = self.in_block(a)
        x2 = _run_block(self.down1, x1)
        x3 = _run_block(self.down2, x2)
        x4 = _run_block(self.down3, x3)
        x5 = _run_block(self.down4, x4)
        u = _dropout(_run_block(self.up2, x5), dropout)
        u = _run_block(self.up3, u)
        # skips (joint_model.py:380,382).  x3 / x2 also feed down3 / down2's strided conv: their skip gradient is parked for that conv's
        # backward to sum in (ops._park_gradient) instead of an add launch of autograd's
        u = _dropout(Act(ops.Materialize.apply(u.raw, u.stats, x3.raw, x3.stats, True), None), dropout)
        u = _run_block(self.up4, u)
        u = _dropout(Act(ops.Materialize.apply(u.raw, u.stats, x2.raw, x2.stats, True), None), dropout)
        u = _dropout(_run_block(self.up5, u), dropout)
        if u.stats is not None:
            ops.mark_defer_apply(u.raw, self.up5.conv[1].conv[6])           # up5's last conv feeds out_block only
        if dropout:     # the reference also drops the two logits before the softmax (joint_model.py:386-388): fused epilogue
            data_dict[out_key] = ops.out_block_softmax(u.raw, u.stats, self.out_block.weight, self.out_block.bias,
                                                       float(dropout), ops.next_dropout_seed())
        else:
            data_dict[out_key] = ops.out_block_softmax(u.raw, u.stats, self.out_block.weight, self.out_block.bias)
        return data_dict


class Joint(nn.Module):
    """joint_model.py:438-452."""

    def __init__(self, models, vae_forward_scale=0.0, vae_decoder_dropout=0.0, seg_dropout=0.0):
        super().__init__()
        self.Seg = models[0]
        self.Vae = models[1]
        self.vae_forward_scale = vae_forward_scale
        self.vae_decoder_dropout = vae_decoder_dropout
        self.seg_dropout = seg_dropout

    def forward(self, data_dict, in_key, out_key, out_key_recon, dropout=False):
        with ops.arena_scope(data_dict[in_key].device):          # Seg and Vae share one statistics arena: one zero fill per forward
            if dropout:
                data_dict = self.Seg(data_dict, in_key, out_key, dropout=self.seg_dropout)
                data_dict[out_key_recon], _, _ = self.Vae(data_dict[out_key], if_random=False, scale=self.vae_forward_scale,
                                                          dropout=self.vae_decoder_dropout)
            else:
                data_dict = self.Seg(data_dict, in_key, out_key)
                data_dict[out_key_recon], data_dict["mean"], data_dict["std"] = self.Vae(
                    data_dict[out_key], if_random=False, scale=self.vae_forward_scale)
        return data_dict


def _linear(x2d, fc, relu):
    """nn.Linear on a (B, K) fp32 tensor through the native GEMV (LinearCL with a 1x1x1 grid: identity flatten order)."""
    return ops.LinearCL.apply(x2d.contiguous().view(x2d.shape[0], 1, 1, 1, x2d.shape[1]), fc.weight, fc.bias, relu)


class Encoder(nn.Module):
    """joint_model.py:274-303 — VAE encoder trunk + fc1 / fc2 / fc_mean with a sigmoid output: the discriminator `Dis` of
    domain_adaptation_dis (main_target.py:338-341) and the image encoder of Embed.  ``spatial`` as in VAE (reference: 128)."""

    def __init__(self, n_channels, dim, norm_type=2, n_fmaps=[8, 16, 32, 64, 128, 256], soft=False, spatial=128):
        super().__init__()
        if spatial % 32 or spatial < 64:
            raise ValueError("spatial must be a multiple of 32 and >= 64")
        f = list(n_fmaps)
        self.in_block = Conv(n_channels, f[0], norm_type=norm_type, soft=False)
        self.down1 = Down(f[0], f[1], norm_type=norm_type, soft=False)
        self.down2 = Down(f[1], f[2], norm_type=norm_type, soft=False)
        self.down3 = Down(f[2], f[3], norm_type=norm_type, soft=False)
        self.down4 = Down(f[3], f[4], norm_type=norm_type, soft=False)
        self.down5 = Down(f[4], f[5], norm_type=norm_type, soft=False)
        self.fc1 = nn.Linear(f[5] * (spatial // 32) ** 3, 1024)
        self.fc2 = nn.Linear(1024, 128)
        self.fc_mean = nn.Linear(128, dim)
        self.spatial = spatial
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, x):
        ops._require_cuda(x)
        if x.shape[-1] != self.spatial:
            raise ValueError("Encoder built for spatial=%d got input side %d" % (self.spatial, x.shape[-1]))
        ops.stats_arena_begin(x.device)
        a = Act(ops.PackPlanar.apply(x.contiguous(), self.kernel_dtype), None)
        a = self.in_block(a)
        for blk in (self.down1, self.down2, self.down3, self.down4, self.down5):
            a = blk(a)
        feat = ops.Materialize.apply(a.raw, a.stats, None, None)
        h = ops.LinearCL.apply(feat, self.fc1.weight, self.fc1.bias, True)
        h = _linear(h, self.fc2, True)
        return torch.sigmoid(_linear(h, self.fc_mean, False))           # (B, dim) scalars: host-side glue like the loss combines


class Fusion(nn.Module):
    """joint_model.py:392-437 — U-Net over an image and a mask branch, added at half resolution, additive skips at up3 / up4."""

    def __init__(self, n_channels_img, n_channels_mask, n_class, norm_type=2, n_fmaps=[8, 16, 32, 64, 128, 256]):
        super().__init__()
        _check_n_class(n_class)
        f = list(n_fmaps)
        self.in_block = Conv(n_channels_img, f[0], norm_type=norm_type, soft=False)
        self.down1 = Down(f[0], f[1], norm_type=norm_type, soft=False)
        self.in_block_mask = Conv(n_channels_mask, f[0], norm_type=norm_type, soft=False)
        self.down1_mask = Down(f[0], f[1], norm_type=norm_type, soft=False)
        self.merge = Conv(f[1], f[1], norm_type=norm_type, soft=False)
        self.down2 = Down(f[1], f[2], norm_type=norm_type, soft=False)
        self.down3 = Down(f[2], f[3], norm_type=norm_type, soft=False)
        self.down4 = Down(f[3], f[4], norm_type=norm_type, soft=False)
        self.up2 = Up(f[4], f[3], norm_type=norm_type, soft=False)
        self.up3 = Up(f[3], f[2], norm_type=norm_type, soft=False)
        self.up4 = Up(f[2], f[1], norm_type=norm_type, soft=False)
        self.up5 = Up(f[1], f[0], norm_type=norm_type, soft=False)
        self.out_block = nn.Conv3d(f[0], n_class, 3, padding=1)
        self.final = nn.Softmax(dim=1)
        self.n_class = n_class
        self.kernel_dtype = _DEFAULT_DTYPE

    def forward(self, data_dict, in_key_img, in_key_mask, out_key):
        x_img, x_mask = data_dict[in_key_img], data_dict[in_key_mask]
        ops._require_cuda(x_img, x_mask)
        if any(s % 16 for s in x_img.shape[2:]):
            raise ValueError("Fusion needs spatial sizes that are multiples of 16, got %s" % (tuple(x_img.shape[2:]),))
        ops.stats_arena_begin(x_img.device)
        a_img = self.down1(self.in_block(Act(ops.PackPlanar.apply(x_img.contiguous(), self.kernel_dtype), None)))
        a_mask = self.down1_mask(self.in_block_mask(Act(ops.PackPlanar.apply(x_mask.contiguous(), self.kernel_dtype), None)))
        x2 = self.merge(Act(ops.Materialize.apply(a_img.raw, a_img.stats, a_mask.raw, a_mask.stats), None))
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        u = self.up3(self.up2(x5))
        u = self.up4(Act(ops.Materialize.apply(u.raw, u.stats, x3.raw, x3.stats), None))
        u = self.up5(Act(ops.Materialize.apply(u.raw, u.stats, x2.raw, x2.stats), None))
        if u.stats is not None:
            ops.mark_defer_apply(u.raw, self.up5.conv[1].conv[6])
        data_dict[out_key] = ops.ConvK3Softmax.apply(u.raw, u.stats, self.out_block.weight, self.out_block.bias)
        return data_dict


class Joint2(nn.Module):
    """joint_model.py:454-466 — segmenter + discriminator on the foreground probability."""

    def __init__(self, models, seg_dropout=0.0):
        super().__init__()
        self.Seg = models[0]
        self.Dis = models[1]
        self.seg_dropout = seg_dropout

    def forward(self, data_dict, in_key, out_key, score_key, dropout=False):
        if dropout:
            data_dict = self.Seg(data_dict, in_key, out_key, dropout=self.seg_dropout)
        else:
            data_dict = self.Seg(data_dict, in_key, out_key)
        data_dict[score_key] = self.Dis(data_dict[out_key][:, 1:2, :, :, :])
        return data_dict


class Embed(nn.Module):
    """joint_model.py:469-500 — image encoder -> latent code -> VAE decoder (initial segmentation) -> Fusion refinement.
    ``noise`` (optional) replaces the VAE's own draw for the `gt_recon` pass, as in VAE.forward."""

    def __init__(self, models):
        super().__init__()
        self.Encoder = models[0]
        self.Vae = models[1]
        self.Fusion = models[2]

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
