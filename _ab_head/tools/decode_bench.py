#!/usr/bin/env python3
"""BASELINE.json configs[4]: CTC decode over a ChaLearn-2013-test-shaped synthetic set (N=276 sequences, T=1900, C=22):
fusion-network inference (learning phase 0) -> thresholded best-path decode (the reference's decoder,
multimodal_fusion/sequence_decoding.py:21-69) and beam search (beam=10) on the GPU; label error rate of the GPU
hypotheses against the CPU oracle's on a bounded sample.  Prints one JSON line.

    python tools/decode_bench.py [--n 276] [--cpu-n 6]

The posteriors fed to the decoders are run-structured (peaky, blank-dominated) like a trained CTC network's; a
random-weight network's own posteriors are near-uniform and make a meaningless decode workload.  The network forward is
timed on its own (it is the `predict_generator` half of the decode script).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mgr_amd  # noqa: E402,F401  (before numpy: _hostenv.py)
import numpy as np  # noqa: E402


def peaky_posteriors(N, T, C, seed):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((N, T, C)).astype(np.float32) * 1.5
    z[:, :, C - 1] += 3.0
    truth = []
    for n in range(N):
        t, seq = 20, []
        while t < T - 40:
            c = int(rng.integers(0, C - 1))
            run = int(rng.integers(5, 40))
            z[n, t:t + run, c] += rng.uniform(3.0, 9.0)
            seq.append(c)
            t += run + int(rng.integers(10, 90))
        truth.append(seq)
    P = np.exp(z - z.max(-1, keepdims=True))
    return (P / P.sum(-1, keepdims=True)).astype(np.float32), truth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=276)
    ap.add_argument("--maxlen", type=int, default=1900)
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--cpu-n", type=int, default=6)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="sequences per inference batch")
    args = ap.parse_args()
    import mgr_amd  # noqa: F401
    from mgr_amd import _capi, decoding
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    from oracle import keras_ref as kr

    dev = _capi.Device(0)
    N, T, C = args.n, args.maxlen, 22
    # ---- predict_generator half: fusion network inference, one batch at a time (round 2's path) and pipelined (Engine.predict_stream)
    spec = fusion_spec()
    Bp = args.batch
    eng = Engine(spec, Bp, T, 1, device=dev, seed=1, inference_only=True)
    eng.set_weights(synthetic_weights(spec, 20131900 + 5))
    nchunks = (N + Bp - 1) // Bp
    chunks = [synthetic_arrays(spec, Bp, T, 1, 20131900 + 5 + i)[0] for i in range(min(nchunks, 3))]
    feed = lambda: (chunks[i % len(chunks)] for i in range(nchunks))
    eng.predict(chunks[0])
    dev.sync()
    t0 = time.perf_counter()
    seq = [eng.predict(x) for x in feed()]
    dev.sync()
    t_pred = time.perf_counter() - t0
    list(eng.predict_stream(feed(), output="posteriors"))            # warm-up (pinned buffers)
    t0 = time.perf_counter()
    pipe = list(eng.predict_stream(feed(), output="posteriors"))
    t_pipe = time.perf_counter() - t0
    same = all(np.array_equal(a, b) for a, b in zip(seq, pipe))
    list(eng.predict_stream(feed(), output="argmax"))
    t0 = time.perf_counter()
    am = list(eng.predict_stream(feed(), output="argmax"))
    ids = [decoding.greedy_decode_argmax(b, p, 0.5) for b, p in am]
    t_e2e_best = time.perf_counter() - t0
    list(eng.predict_stream(feed(), output="beam", beam_width=args.beam))
    t0 = time.perf_counter()
    bm = list(eng.predict_stream(feed(), output="beam", beam_width=args.beam))
    t_e2e_beam = time.perf_counter() - t0
    eng.close()

    # ---- decode half
    P, truth = peaky_posteriors(N, T, C, 20131900 + 5)
    il = np.full(N, T - 2)
    decoding.greedy_decode(P[:4], 0.5, dev=dev)
    decoding.beam_search_decode(P[:4], il[:4], beam_width=args.beam, dev=dev)
    tg = tb = 1e30
    for _ in range(args.reps):
        t0 = time.perf_counter()
        greedy = decoding.greedy_decode(P, 0.5, dev=dev)
        tg = min(tg, time.perf_counter() - t0)
        t0 = time.perf_counter()
        beam, scores = decoding.beam_search_decode(P, il, beam_width=args.beam, dev=dev)
        tb = min(tb, time.perf_counter() - t0)
    # ---- CPU oracle on a bounded sample
    k = min(args.cpu_n, N)
    t0 = time.perf_counter()
    ref_beam, ref_scores = kr.ctc_beam_search(P[:k], il[:k], beam_width=args.beam)
    t_cpu_beam = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_greedy = kr.greedy_decode_quirk(P[:k], 0.5)
    t_cpu_greedy = time.perf_counter() - t0
    out = {
        "metric": "CTC decode, ChaLearn-2013-test-shaped synthetic set", "n_sequences": N, "maxlen": T, "classes": C,
        "beam_width": args.beam,
        "predict_batch": Bp, "predict_batches": nchunks,
        "predict_one_batch_at_a_time": {"sequences_per_s": round(nchunks * Bp / t_pred, 1), "ms_per_batch": round(t_pred / nchunks * 1e3, 2),
                                        "frames_per_s": round(nchunks * Bp * T / t_pred)},
        "predict_pipelined": {"sequences_per_s": round(nchunks * Bp / t_pipe, 1), "ms_per_batch": round(t_pipe / nchunks * 1e3, 2),
                              "frames_per_s": round(nchunks * Bp * T / t_pipe), "bit_identical_to_sequential": bool(same),
                              "incl": "host batch -> pinned staging -> H2D, (B,T,C) posteriors -> host"},
        "end_to_end_pipelined": {"predict_plus_best_path": {"sequences_per_s": round(nchunks * Bp / t_e2e_best, 1), "ms_total": round(t_e2e_best * 1e3, 1),
                                                            "note": "frame arg-max on the device, threshold filter + collapse on the host"},
                                 "predict_plus_beam": {"sequences_per_s": round(nchunks * Bp / t_e2e_beam, 1), "ms_total": round(t_e2e_beam * 1e3, 1),
                                                       "note": "beam search on the device from the posteriors in HBM (its own stream, beside the next batch).  "
                                                               "WORST CASE for the beam kernel: a random-weight network's posteriors are near-uniform, every "
                                                               "frame extends every prefix; beam_sequences_per_s below is the rate on peaky, trained-like posteriors"}},
        "beam_sequences_per_s": round(N / tb, 1), "beam_ms_total_incl_h2d_d2h": round(tb * 1e3, 2),
        "best_path_sequences_per_s": round(N / tg, 1), "best_path_ms_total_incl_h2d_d2h": round(tg * 1e3, 2),
        "cpu_oracle": {"sample": k, "beam_sequences_per_s": round(k / t_cpu_beam, 3),
                       "best_path_sequences_per_s": round(k / t_cpu_greedy, 1), "cores": 1, "kind": "port"},
        "cpu_oracle_comparison_covers": "%d of %d sequences (all %d are compared in tests/test_gpu_fullsize.py::test_config_D_decode_full_set_matches_oracle_on_every_sequence)" % (k, N, N),
        "label_error_rate_vs_cpu_ref": {"beam": decoding.label_error_rate(beam[:k], ref_beam),
                                        "best_path": decoding.label_error_rate(greedy[:k], ref_greedy)},
        "exact_match_vs_cpu_ref": {"beam": sum(a == b for a, b in zip(beam[:k], ref_beam)) / k,
                                   "best_path": sum(a == b for a, b in zip(greedy[:k], ref_greedy)) / k},
        "beam_score_max_rel_delta": float(np.max(np.abs((scores[:k] - ref_scores) / ref_scores))),
    }
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
