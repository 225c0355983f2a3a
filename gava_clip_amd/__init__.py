"""gava_clip_amd — MI355X (gfx950) implementation of the GaVA-CLIP video-frame forward path.

``from gava_clip_amd import VitaCLIP`` is the drop-in for
``from VitaCLIP_model import VitaCLIP`` (/root/reference/training/train.py:30).
"""
from .config import VitaConfig, VIT_B16_T8, VIT_B16_T16, VIT_L14_T32, TINY, param_shapes  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch/HIP
    if name == "VitaCLIP":
        from .model import VitaCLIP
        return VitaCLIP
    raise AttributeError(name)
