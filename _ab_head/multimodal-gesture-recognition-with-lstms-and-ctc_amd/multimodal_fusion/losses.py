"""``ctc_lambda_func`` (reference multimodal_fusion/losses.py:4-15), evaluated by the HIP CTC kernel."""
import numpy as np

from .. import _capi
from ..decoding import default_device


def ctc_lambda_func(args, skip=2, eps=1e-8, dev=None):
    """args = [y_pred (B,T,C) softmax, labels (B,Lmax) padded -1, input_length (B,1), label_length (B,1)].

    Drops the first ``skip`` frames ("the first couple outputs of the RNN tend to be garbage") and returns the
    per-sample CTC cost, shape (B,1), exactly what K.ctc_batch_cost yields inside the reference's Lambda layer."""
    y_pred, labels, input_length, label_length = args
    dev = dev or default_device()
    P = np.ascontiguousarray(y_pred, dtype=np.float32)
    B, T, C = P.shape
    lab = np.asarray(labels)
    lab = np.where(np.isfinite(lab), lab, -1).astype(np.int32).reshape(B, -1)
    Lmax = lab.shape[1]
    dP = dev.array(P)
    dl = dev.array(lab)
    dil = dev.array(np.asarray(input_length).reshape(B).astype(np.int32))
    dll = dev.array(np.asarray(label_length).reshape(B).astype(np.int32))
    loss = dev.empty((B,))
    ws = dev.bytes(dev.lib.mgr_ctc_ws_bytes(B, T, C, Lmax))
    dev.call("mgr_ctc_loss_grad", dP, dl, dil, dll, B, T, C, Lmax, skip, C - 1, eps, 1.0, loss, 0, ws, ws.nbytes)
    out = loss.download().reshape(B, 1)
    for a in (dP, dl, dil, dll, loss, ws):
        a.free()
    return out
