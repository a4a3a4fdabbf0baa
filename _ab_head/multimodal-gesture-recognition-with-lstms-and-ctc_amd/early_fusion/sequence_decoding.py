"""Decode for the early-fusion network (reference early_fusion/sequence_decoding.py:261-306): best path with a
0.97 confidence threshold, the list.remove quirk, collapse, MLF."""
import numpy as np

from ..decoding import greedy_decode, write_mlf
from ..multimodal_fusion.sequence_decoding import ignore_list, map_gest

THRESHOLD = 0.97


def decode_batch(pred_out, f_list, out_file="final_ctc_recout.mlf"):
    ids = greedy_decode(np.asarray(pred_out), THRESHOLD, skip=2)
    ret = [[map_gest[i] for i in seq] for seq in ids]
    write_mlf(out_file, ret, f_list, ignore_list, "Sample%05d")
    return ret
