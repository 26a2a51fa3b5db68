"""MI355X-native drop-in for the U-Net hot path of NIH-NEI/oct-image-segmentation-models.

Mirrors the reference package layout (``models``, ``common``, ``training``,
``evaluation``, ``prediction``, ``min_path_processing``) for the parts on or
next to the hot path; the arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/oct_unet.h`` (``liboct_unet_hip.so``).  There is no CPU
fallback: the engine raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
