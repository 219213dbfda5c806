"""Drop-in for the reference's joint_model.py (loaded by name: main_source.py:247, main_target.py:314).

Same class names, constructor / forward signatures and state_dict keys; the arithmetic runs on the
hand-written gfx950 kernels of vae_segmentation_amd (libvaeseg.so).  GPU only — no CPU fallback."""
from vae_segmentation_amd.modules import (Conv, DoubleConv, Down, Embed, Encoder, Fusion, Joint, Joint2, Normalization,  # noqa: F401
                                          Segmentation, Up, VAE,
                                          set_default_kernel_dtype, set_kernel_dtype, set_recompute)
from vae_segmentation_amd.modules_gs import (Conv_GS, DoubleConv_GS, Down_GS, GSConv3d, GSConvTranspose3d, GSNorm3d, SConv3d,  # noqa: F401
                                             Segmentation_GS, Up_GS)
