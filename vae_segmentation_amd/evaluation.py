"""Loss / metric functions of the reference's utils/evaluation.py (and their duplicates in
main_source.py:133-182) on the libvaeseg kernels.  Same names, arguments and dict-in conventions;
``eps`` selects between the two epsilons that coexist in the reference:
utils/evaluation.py uses 1e-6, main_source.py's own copy 1e-4."""
import torch

from . import ops

EPS_EVALUATION = 1e-6
EPS_MAIN_SOURCE = 1e-4


def dice(A, B):
    """utils/evaluation.py:6-7 — whole-tensor soft dice (eps 1e-6)."""
    a = A.reshape(1, 1, -1)
    b = B.reshape(1, 1, -1)
    pad = (-a.shape[-1]) % 4
    if pad:
        a = torch.nn.functional.pad(a, (0, pad))
        b = torch.nn.functional.pad(b, (0, pad))
    return ops.Dice.apply(a, b, 0, 1, EPS_EVALUATION, True)


def binarize(A):
    """utils/evaluation.py:9-10."""
    return ops.binarize(A, mode=0)


def confident_binarize(A, max=0.8, min=0.2):
    """utils/evaluation.py:12-18."""
    return ops.binarize(A, mode=1, lo=min, hi=max)


def avg_ce(data_dict, source_key='align_lung', target_key='source_lung'):
    """utils/evaluation.py:29-39."""
    source_mask = data_dict[source_key]
    target_mask = data_dict[target_key]
    if not isinstance(source_mask, list):
        source_mask = [source_mask]
    total = 0
    for im in source_mask:
        total = total + ops.BCE.apply(im, target_mask)
    return total / len(source_mask)


def KLloss(data_dict, mean_key='mean', std_key='std'):
    """utils/evaluation.py:42-45."""
    return ops.KL.apply(data_dict[mean_key], data_dict[std_key])


def _hard_onehot(mask):
    """argmax over channels -> one-hot (validation path, utils/evaluation.py:58-64), any number of classes: vs_hard_onehot
    (ties go to the first maximal channel, as torch.argmax resolves them)."""
    return ops.hard_onehot(mask)


def avg_dsc(data_dict, source_key='align_lung', target_key='source_lung', binary=False, topindex=2, botindex=0,
            pad=[0, 0, 0], return_mean=True, detach=False, eps=EPS_EVALUATION):
    """utils/evaluation.py:48-80 (eps=1e-6) / main_source.py:150-182 (eps=1e-4)."""
    source_mask = data_dict[source_key]
    target_mask = data_dict[target_key]
    if detach:
        target_mask = target_mask.detach()
    if binary:
        source_mask = _hard_onehot(source_mask)
        target_mask = _hard_onehot(target_mask)
    channels = source_mask.shape[1]
    if channels > 1:
        bot, top = botindex, min(topindex, channels)
    else:
        bot, top = 0, 1
    return ops.Dice.apply(source_mask, target_mask, bot, top, eps, return_mean)
