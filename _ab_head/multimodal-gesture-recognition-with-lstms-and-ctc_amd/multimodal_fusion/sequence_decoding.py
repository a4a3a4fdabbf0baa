"""Decode for the fusion network (reference multimodal_fusion/sequence_decoding.py:21-69)."""
import numpy as np

from ..decoding import greedy_decode, greedy_decode_argmax, write_mlf

# gesture code -> class name; the blank (21) is emitted as "sil" (reference :26-29)
map_gest = {0: "oov", 1: "VA", 2: "VQ", 3: "PF", 4: "FU", 5: "CP", 6: "CV", 7: "DC", 8: "SP", 9: "CN", 10: "FN",
            11: "OK", 12: "CF", 13: "BS", 14: "PR", 15: "NU", 16: "FM", 17: "TT", 18: "BN", 19: "MC", 20: "ST",
            21: "sil"}
# files the reference skips when writing the MLF (:32)
ignore_list = [228, 298, 299, 300, 303, 304, 334, 343, 373, 375]
THRESHOLD = 0.5


def decode_batch(pred_out, f_list, out_file="final_ctc_recout.mlf"):
    """pred_out (N,T,C) softmax, f_list file numbers.  Writes the MLF and returns the label-name lists
    (ignored files included in the return value, as in the reference)."""
    ids = greedy_decode(np.asarray(pred_out), THRESHOLD, skip=2)
    ret = [[map_gest[i] for i in seq] for seq in ids]
    write_mlf(out_file, ret, f_list, ignore_list, "Sample%05d")
    return ret


def decode_argmax(best, prob, f_list, out_file=None):
    """decode_batch from the per-frame (best label, probability) pairs that Model.predict_generator(decode="argmax") computes
    on the device: same filter, same collapse, same MLF."""
    ids = greedy_decode_argmax(best, prob, THRESHOLD)
    ret = [[map_gest[i] for i in seq] for seq in ids]
    if out_file is not None:
        write_mlf(out_file, ret, f_list, ignore_list, "Sample%05d")
    return ret
