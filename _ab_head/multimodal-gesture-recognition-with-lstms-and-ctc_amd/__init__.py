"""mgr_amd - MI355X-native BiLSTM + CTC training / decode path behind the reference's Python surface.

Layout: csrc/ (HIP kernels + C ABI, built to libmgr.so), _capi.py (ctypes), engine.py (device sequencing),
keras_like.py (Keras-shaped façade), and the reference's three script packages
(audio_network, skeletal_network, multimodal_fusion) with the same module and symbol names.
"""
from ._hostenv import bound_thread_pools

bound_thread_pools()   # before anything below imports numpy

from .spec import NetworkSpec  # noqa: E402,F401

__all__ = ["NetworkSpec"]
__version__ = "0.1.0"
