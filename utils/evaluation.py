"""Drop-in for the reference's utils/evaluation.py (imported by main_target.py:23)."""
from vae_segmentation_amd.evaluation import KLloss, avg_ce, avg_dsc, binarize, confident_binarize, dice  # noqa: F401
