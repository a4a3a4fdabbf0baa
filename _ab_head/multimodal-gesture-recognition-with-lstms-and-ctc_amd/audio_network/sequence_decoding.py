"""Decode for the audio network (reference audio_network/sequence_decoding.py:19-69): word-level classes, thr .75."""
import numpy as np

from ..decoding import greedy_decode, greedy_decode_argmax, write_mlf

_words = ["oov", "Vattene", "Vieni", "qui", "Perfetto", "E'", "un", "furbo", "Che", "due", "palle", "vuoi", "Vanno",
          "d'accordo", "Sei", "Pazzo", "Cos'hai", "combinato", "Non", "me", "ne", "frega", "niente", "ok", "Cosa", "ti",
          "farei", "Basta", "Le", "prendere", "ce", "n'e", "piu", "Ho", "fame", "Tanto", "tempo", "fa", "Buonissimo",
          "Si", "sono", "messi", "stufo", "sil"]
map_gest = dict(enumerate(_words))
map_gest[-1] = "sil"
ignore_list = [228, 298, 299, 300, 303, 304, 334, 343, 373, 375]
THRESHOLD = 0.75


def decode_batch(pred_out, f_list, out_file="ctc_recout.mlf"):
    ids = greedy_decode(np.asarray(pred_out), THRESHOLD, skip=2)
    ret = [[map_gest[i] for i in seq] for seq in ids]
    write_mlf(out_file, ret, f_list, ignore_list, "Sample%05d_audio")
    return ret


def decode_argmax(best, prob, f_list, out_file=None):
    """decode_batch from the per-frame (best label, probability) pairs that Model.predict_generator(decode="argmax") computes
    on the device: same filter, same collapse, same MLF."""
    ids = greedy_decode_argmax(best, prob, THRESHOLD)
    ret = [[map_gest[i] for i in seq] for seq in ids]
    if out_file is not None:
        write_mlf(out_file, ret, f_list, ignore_list, "Sample%05d_audio")
    return ret
