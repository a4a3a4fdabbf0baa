"""``ctc_lambda_func`` (reference audio_network/losses.py:4-15 - a verbatim copy of multimodal_fusion/losses.py there; here one
implementation, re-exported)."""
from ..multimodal_fusion.losses import ctc_lambda_func  # noqa: F401
