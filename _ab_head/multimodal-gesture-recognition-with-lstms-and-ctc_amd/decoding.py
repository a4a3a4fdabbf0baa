"""Decode helpers shared by the three sequence_decoding modules (K9).

Frame-wise max / argmax runs on the GPU (mgr_frame_argmax); the reference's confidence filter and the
repeat collapse run on the host.  The filter reproduces the NET EFFECT of the reference's Python-2 loop
(multimodal_fusion/sequence_decoding.py:45-48: ``list.remove`` deletes the first element equal to the value,
not the visited one): for every label s, the first k_s occurrences of s are dropped, where k_s is the number
of frames whose best label is s with probability below the threshold.  Blanks are kept (they decode to "sil").
"""
import ctypes as C

import numpy as np

from . import _capi

_DEV = [None]


def default_device():
    if _DEV[0] is None:
        _DEV[0] = _capi.Device(0)
    return _DEV[0]


def frame_argmax(pred_out, skip=2, dev=None):
    """(N,T,C) float32 softmax -> best (N,T-skip) int32, prob (N,T-skip) float32, computed on the GPU."""
    dev = dev or default_device()
    P = np.ascontiguousarray(pred_out, dtype=np.float32)
    N, T, Cn = P.shape
    dP = dev.array(P)
    best = dev.empty((N, T - skip), np.int32)
    prob = dev.empty((N, T - skip), np.float32)
    dev.call("mgr_frame_argmax", dP, N, T, Cn, skip, best, prob)
    out = best.download(), prob.download()
    for a in (dP, best, prob):
        a.free()
    return out


def confidence_filter_collapse(best, prob, thr):
    """One sample: best/prob 1-D arrays over frames.  Returns the collapsed label-id list."""
    best = np.asarray(best)
    if thr is not None:
        low = np.asarray(prob) < thr          # float32 value vs Python float, as in the reference
        nlab = int(best.max()) + 1 if best.size else 0
        k = np.bincount(best[low], minlength=nlab)
        # occurrence index of every frame among the frames with the same label
        order = np.argsort(best, kind="stable")
        sorted_lab = best[order]
        starts = np.r_[0, np.flatnonzero(np.diff(sorted_lab)) + 1]
        group_start = np.repeat(starts, np.diff(np.r_[starts, best.size]))
        rank = np.empty(best.size, np.int64)
        rank[order] = np.arange(best.size) - group_start
        best = best[rank >= k[best]]
    if best.size == 0:
        return []
    keep = np.r_[True, best[1:] != best[:-1]]
    return [int(v) for v in best[keep]]


def greedy_decode(pred_out, thr, skip=2, dev=None):
    best, prob = frame_argmax(pred_out, skip, dev)
    return [confidence_filter_collapse(best[j], prob[j], thr) for j in range(best.shape[0])]


def greedy_decode_argmax(best, prob, thr):
    """The same from per-frame (best label, its probability) arrays - what Model.predict_generator(decode="argmax") returns
    straight from the device, without the (N, T, C) posteriors crossing PCIe."""
    best, prob = np.asarray(best), np.asarray(prob)
    return [confidence_filter_collapse(best[j], prob[j], thr) for j in range(best.shape[0])]


def write_mlf(path, decoded_names, f_list, ignore_list, name_fmt="Sample%05d"):
    """HTK master label file in the reference's layout (sequence_decoding.py:35-36,57-65)."""
    with open(path, "w") as of:
        of.write("#!MLF!#\n")
        for names, f_num in zip(decoded_names, f_list):
            if int(f_num) in ignore_list:
                continue
            of.write('"*/%s.rec"\n' % (name_fmt % int(f_num)))
            for cl in names:
                of.write("%s\n" % cl)
            of.write(".\n")


def beam_search_decode(pred_out, input_length=None, beam_width=10, skip=2, merge_repeated=True, dev=None):
    """K.ctc_decode(greedy=False, beam_width) equivalent on the GPU (BASELINE.json config 5)."""
    dev = dev or default_device()
    P = np.ascontiguousarray(pred_out, dtype=np.float32)
    N, T, Cn = P.shape
    if input_length is None:
        input_length = np.full(N, T - skip)
    il = np.asarray(input_length).reshape(N).astype(np.int32)
    dP, dil = dev.array(P), dev.array(il)
    out = dev.empty((N, T - skip), np.int32)
    olen = dev.empty((N,), np.int32)
    logp = dev.empty((N,), np.float64)
    ws = dev.bytes(dev.lib.mgr_ctc_beam_ws_bytes(N, T, Cn, beam_width))
    dev.call("mgr_ctc_beam_search", dP, dil, N, T, Cn, skip, Cn - 1, int(beam_width), C.c_float(1e-8),
             1 if merge_repeated else 0, out, olen, logp, ws, ws.nbytes)
    o, l, s = out.download(), olen.download(), logp.download()
    for a in (dP, dil, out, olen, logp, ws):
        a.free()
    return [[int(v) for v in o[i, :l[i]]] for i in range(N)], s


def edit_distance(a, b):
    la, lb = len(a), len(b)
    d = list(range(lb + 1))
    for i in range(1, la + 1):
        prev, d[0] = d[0], i
        for j in range(1, lb + 1):
            cur = d[j]
            d[j] = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev = cur
    return d[lb]


def label_error_rate(hyps, refs):
    """sum of edit distances / sum of reference lengths (what HTK HResults reports as 100 - Acc)."""
    num = sum(edit_distance(h, r) for h, r in zip(hyps, refs))
    den = sum(len(r) for r in refs)
    return num / max(1, den)


def read_mlf(path):
    """HTK master label file -> {sample name: [labels]} (the layout write_mlf produces: '"*/Sample00001.rec"' lines,
    one label per line, '.' terminator).  '.lab' and '.rec' entries are keyed by the bare sample name."""
    out, cur = {}, None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line == "#!MLF!#":
                continue
            if line.startswith('"'):
                name = line.strip('"').split("/")[-1]
                cur = out.setdefault(name.rsplit(".", 1)[0], [])
            elif line == ".":
                cur = None
            elif cur is not None:
                cur.append(line.split()[-1] if len(line.split()) > 1 and line.split()[0].isdigit() else line.split()[0])
    return out


def score_mlf(ref_path, rec_path, ignore=("sil",)):
    """Label error rate of a recognition MLF against a reference MLF, over the samples present in both
    (HResults-style: (S + D + I) / N after removing the `ignore` labels).  Returns (ler, n_samples)."""
    ref, rec = read_mlf(ref_path), read_mlf(rec_path)
    names = sorted(set(ref) & set(rec))
    strip = lambda seq: [x for x in seq if x not in ignore]
    return label_error_rate([strip(rec[n]) for n in names], [strip(ref[n]) for n in names]), len(names)
