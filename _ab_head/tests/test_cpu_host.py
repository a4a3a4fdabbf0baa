"""CPU suite (no GPU): host logic, the C-ABI surface, and the fail-loud behaviour without a GPU."""
import io
import time
import os
import re

import numpy as np
import pytest

import mgr_amd  # noqa: F401
from mgr_amd import _capi, configs, decoding
from mgr_amd.keras_like import Adam, Model, ModelCheckpoint, model_from_json
from mgr_amd.spec import NetworkSpec
from oracle import keras_ref as kr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mgr.h")).read()
    declared = set(re.findall(r"\b(mgr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mgr_ctx", "mgr_comm", "mgr_scan_job"}
    lib = _capi.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "libmgr.so does not export %s" % name
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert lib.mgr_version() >= 100


def test_no_gpu_fails_loudly():
    if _capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_capi.MgrError):
        _capi.Device(0)
    from mgr_amd.engine import Engine
    with pytest.raises(_capi.MgrError):
        Engine(configs.fusion_spec(h_audio=8, h_skeletal=8, h_fusion=4), 2, 8, 4)


def test_product_never_imports_oracle_or_torch():
    pkg = os.path.join(ROOT, "multimodal-gesture-recognition-with-lstms-and-ctc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f
                assert "import torch" not in src, f


def test_spec_accounting_matches_survey():
    spec, B, T, Lmax = configs.baseline_config("F")
    assert spec.flops_per_frame() == 27794400
    assert spec.count_params(trainable_only=True) == 1365222
    assert spec.count_params() - spec.count_params(True) == 11096800
    assert (B, T, Lmax) == (64, 1900, 35) and spec.concat_width == 1600
    a = configs.audio_spec()
    assert a.count_params() == 8208044
    s = configs.skeletal_spec()
    assert s.count_params() == 2946022
    assert NetworkSpec.from_json(spec.to_json()).to_dict() == spec.to_dict()


def test_filter_collapse_equals_literal_python2_loop():
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        Cn = int(rng.integers(2, 6))
        best = rng.integers(0, Cn, size=n)
        prob = rng.random(n).astype(np.float32)
        P = np.zeros((1, n + 2, Cn), np.float32)
        for t in range(n):
            P[0, t + 2, :] = (1 - prob[t]) / (Cn - 1) * 0.5
            P[0, t + 2, best[t]] = max(prob[t], 0.51)  # keep argmax == best
        p_eff = P[0, 2:].max(-1)
        thr = 0.75
        assert decoding.confidence_filter_collapse(best, p_eff, thr) == kr.greedy_decode_quirk(P, thr)[0]


def test_data_generator_contract(tmp_path):
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    g = DataGenerator(4, 20, 39, 60, 22, 'train', synthetic_files=27)
    assert g.get_size(True) % 4 == 0 and g.get_size(False) % 4 == 0
    assert g.get_size(True) == 20 and g.get_size(False) == 4      # int(27*.8)=21 -> 20 ; 6 -> 4
    gen = g.next_train()
    x, y = next(gen)
    assert set(x) == {'the_input_audio', 'the_input_skeletal', 'the_labels', 'input_length', 'label_length'}
    assert x['the_input_audio'].shape == (4, 60, 39) and x['the_input_audio'].dtype == np.float64
    assert x['the_labels'].shape == (4, 35) and np.all(x['input_length'] == 58)
    assert set(y) == {'ctc'} and y['ctc'].shape == (4,)
    for i in range(4):
        L = int(x['label_length'][i, 0])
        assert np.all(x['the_labels'][i, L:] == -1) and np.all(x['the_labels'][i, :L] >= 0)
    # post-padding with zeros after the true length
    assert np.any(np.all(x['the_input_audio'][:, -1, :] == 0, axis=1))
    # wrap-around after an epoch worth of batches
    for _ in range(g.get_size(True) // 4):
        next(gen)
    assert g.train_index == 4
    # validation / final sets are not split
    v = DataGenerator(2, 20, 39, 60, 22, 'final', synthetic_files=5)
    xb, _ = v.get_batch(False)
    assert v.get_size(False) == 5 and np.all(xb['label_length'] == 1) and np.all(xb['the_labels'][:, 0] == 0)


def test_empty_label_row_substitutes_blank():
    from mgr_amd.datagen import SyntheticStore
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    g = DataGenerator(2, 20, 39, 40, 22, 'val', synthetic_files=4)
    g.store = SyntheticStore(4, {'audio': (39, 3.0), 'skeletal': (20, 1.0)}, 40, 22, empty_every=2)
    x, _ = g.get_batch(False)   # files 1, 2: file 2 has no labels
    assert x['label_length'][1, 0] == 1 and x['the_labels'][1, 0] == 21 and np.all(x['the_labels'][1, 1:] == -1)
    assert np.all(x['the_input_audio'][1] == 1.0)          # inputs stay all-ones, like the reference
    assert not np.all(x['the_input_audio'][0] == 1.0)


def test_audio_word_expansion_and_csv_backend(tmp_path):
    import pandas as pd
    from mgr_amd.audio_network.data_generator import DataGenerator, class_2_words
    g = DataGenerator(2, 39, 50, 44, 'val', synthetic_files=4)
    assert list(g.sent_2_words([2, 10])) == [2, 3, 18, 19, 20, 21, 22]
    assert max(max(v) for v in class_2_words.values()) == 43
    # CSV layout of util/mix_data.py: per-file audio CSVs (100 fps -> every 5th row), label CSV Id/Sequence
    root = tmp_path / "data"
    (root / "val_audio").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for fid, n in ((3, 57), (7, 31)):
        df = pd.DataFrame(rng.standard_normal((n, 39)), columns=[str(i) for i in range(39)])
        df['file_number'] = fid
        df.to_csv(root / "val_audio" / ("audio_%d.csv" % fid), index=False)
    pd.DataFrame({"Id": [3, 7], "Sequence": ["1 2", "20"]}).to_csv(root / "validation.csv", index=False)
    c = DataGenerator(2, 39, 20, 44, 'val', data_root=str(root))
    x, _ = c.get_batch(False)
    assert c.get_file_list(False) == [3, 7]
    assert list(x['the_labels'][0][:4]) == [1, 2, 3, -1] and x['label_length'][1, 0] == 2   # 20 -> [40, 42]
    assert np.all(x['the_input'][0, 12:] == 0) and not np.all(x['the_input'][0, 11] == 0)   # ceil(57/5) = 12 frames


def test_model_facade_host_side(tmp_path):
    spec = configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=4)
    m = Model(spec)
    buf = io.StringIO()
    m.summary(file=buf)
    assert "blstm_2 (Bidirectional)" in buf.getvalue() and "Trainable params" in buf.getvalue()
    assert m.get_layer('softmax').name == 'softmax' and m.layers[0].name == 'the_input_audio'
    enc = m.get_layer('speech_blstm_1')
    assert enc.kind == "Bidirectional" and enc.trainable_weights == []       # frozen encoder
    assert len(m.get_layer('blstm_2').trainable_weights) == 6
    m.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-4, clipvalue=0.5, decay=1e-5))
    assert m.spec.optimizer["decay"] == 1e-5 and m.spec.optimizer["epsilon"] == 1e-7
    p = tmp_path / "w_best.h5"
    m.save_weights(str(p))
    m2 = model_from_json(m.to_json())
    w2 = m2.get_weights_dict()
    w2["dense/b"] = w2["dense/b"] + 1
    m2.set_weights_dict(w2)
    m2.load_weights(str(p))
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    # layer_trainable reproduces the reference's attribute semantics
    from mgr_amd.multimodal_fusion.multimodal import layer_trainable
    layer_trainable(enc, freeze=True)
    assert enc.trainable is True and enc.forward_layer.trainable is False and enc.backward_layer.trainable is False
    # ModelCheckpoint(save_best_only) only writes on improvement
    ck = ModelCheckpoint(str(tmp_path / "best.h5"), monitor='val_loss', save_best_only=True, save_weights_only=True)
    ck.set_model(m)
    ck.on_epoch_end(0, {"val_loss": 2.0})
    t0 = os.path.getmtime(tmp_path / "best.h5")
    ck.on_epoch_end(1, {"val_loss": 3.0})
    assert os.path.getmtime(tmp_path / "best.h5") == t0 and ck.best == 2.0


def test_shard_batch():
    from mgr_amd.parallel import shard_batch
    batch = {"a": np.arange(8).reshape(8, 1), "b": np.arange(16).reshape(8, 2)}
    s1 = shard_batch(batch, 1, 4)
    assert s1["a"].ravel().tolist() == [2, 3] and s1["b"].shape == (2, 2)
    with pytest.raises(ValueError):
        shard_batch(batch, 0, 3)


def test_mlf_round_trip_and_scoring(tmp_path):
    from mgr_amd import decoding
    hyp = [["sil", "VA", "sil", "OK"], ["sil"], ["CP", "CV"]]
    ref = [["VA", "OK"], ["FU"], ["CP", "sil", "CV"]]
    decoding.write_mlf(str(tmp_path / "rec.mlf"), hyp, [1, 2, 3], [], "Sample%05d")
    decoding.write_mlf(str(tmp_path / "ref.mlf"), ref, [1, 2, 3, ], [], "Sample%05d")
    got = decoding.read_mlf(str(tmp_path / "rec.mlf"))
    assert got == {"Sample00001": hyp[0], "Sample00002": hyp[1], "Sample00003": hyp[2]}
    ler, n = decoding.score_mlf(str(tmp_path / "ref.mlf"), str(tmp_path / "rec.mlf"))
    assert n == 3 and ler == pytest.approx(1 / 5)      # one deletion (FU) over 5 reference labels


def test_skeletal_feature_oracle_properties():
    """The CPU restatement of skeletal_feature_extraction.py: shift-over-whole-table and first-five-rows rules."""
    from oracle import skeletal_ref as sr
    rng = np.random.default_rng(0)
    n = 40
    J = {c: rng.uniform(0, 640, n) for c in sr.JOINT_COLS}
    F = sr.extract_features(J)
    assert list(F) == sr.FEATURE_COLS and all(v.shape == (n,) for v in F.values())
    assert np.all(F['lh_v'][:5] == 0) and np.all(F['re_a'][:5] == 0)
    assert F['lh_v'][7] == np.sqrt((J['lhX'][7] - J['lhX'][6]) ** 2 + (J['lhY'][7] - J['lhY'][6]) ** 2)
    assert F['lh_a'][5] == F['lh_v'][5] and F['rh_a'][9] == F['rh_v'][9] - F['rh_v'][8]
    assert F['lh_el_ang'][3] == np.arctan2(J['lhY'][3] - J['leY'][3], J['lhX'][3] - J['leX'][3])
    assert F['re_shc_d'][0] == np.hypot(J['reX'][0] - J['shcX'][0], J['reY'][0] - J['shcY'][0]) or \
        abs(F['re_shc_d'][0] - np.hypot(J['reX'][0] - J['shcX'][0], J['reY'][0] - J['shcY'][0])) < 1e-12


def test_no_register_polling_left_in_the_device_sources():
    """Round 2's default scan step issued its gather loads from inline asm and polled the destination registers - correct only
    as long as hipcc kept each polled value in the registers its load wrote, which a regex over the generated assembly had to
    check at every build.  Round 3's step uses loads the compiler sees and waits for; this pins that the idiom (an asm load with a
    read-write register operand, a build that inspects assembly) does not come back unnoticed."""
    import re
    from mgr_amd import _build
    assert not hasattr(_build, "ISA_CHECKED") and not hasattr(_build, "check_hidden_loads")
    csrc = os.path.join(os.path.dirname(_build.__file__), "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r'asm\s+volatile\s*\(\s*"([^"]*load[^"]*)"((?:[^;]|\n)*?)\);', text):
            assert '"+v"' not in m.group(2), (f, m.group(0)[:120])     # no load into a register the compiler thinks it owns
    assert "MGR_CXXFLAGS" not in open(_build.__file__).read()           # no diagnostic-macro hook in the product build


def test_failed_build_leaves_no_library(tmp_path, monkeypatch):
    """_build.build(): a compiler failure is reported only after EVERY compiler process has ended (none may outlive the build
    and race a retry), and a library of an older build does not survive it - nothing can load a stale libmgr.so."""
    from mgr_amd import _build
    lib = tmp_path / "libmgr.so"
    lib.write_bytes(b"old")
    calls = tmp_path / "calls"
    fake = tmp_path / "hipcc"
    fake.write_text("#!/bin/sh\necho x >> %s\nsleep 0.2\necho 'error: no such thing' >&2\nexit 1\n" % calls)
    fake.chmod(0o755)
    monkeypatch.setattr(_build, "LIB", str(lib))
    monkeypatch.setattr(_build, "OBJDIR", str(tmp_path / "build"))   # objects go to a scratch directory
    monkeypatch.setattr(_build, "_hipcc", lambda: str(fake))
    with pytest.raises(RuntimeError, match="hipcc failed on"):
        _build.build(force=True, verbose=False)
    assert not lib.exists()
    n = len(open(calls).read().split())
    import time
    time.sleep(0.5)
    assert n == len(_build.SOURCES) == len(open(calls).read().split())   # all were started AND had ended when build() raised


def test_train_on_batch_with_fresh_temporaries_never_reuses_a_stale_split():
    """Model._cached_split must match batches by object identity while holding the object: an id() key alone is reused by
    CPython as soon as the previous dict is freed and would make train_on_batch train on the first batch's inputs."""
    spec = configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=4)
    m = Model(spec)

    def make(i):
        return {"the_input_audio": np.full((2, 6, 39), float(i)), "the_input_skeletal": np.full((2, 6, 20), float(-i))}

    for i in range(8):
        ins = m._cached_split(make(i))
        assert ins["the_input_audio"][0, 0, 0] == float(i) and ins["the_input_skeletal"][0, 0, 0] == float(-i)
    x = make(3)
    assert m._cached_split(x) is m._cached_split(x)       # the same object twice: one split (prefetch matching relies on it)


def test_executed_flop_accounting():
    """bench.py reports the algorithmic FLOP of SURVEY 8(d) and, beside it, what the dropout-aware kernels really execute."""
    spec, B, T, Lmax = configs.baseline_config("F")
    alg, ex = spec.flops_per_frame(), spec.flops_per_frame(executed=True)
    assert alg == 27794400 and 0.6 * alg < ex < 0.8 * alg
    # by hand: projections of audio l1 (1000 -> 500, p .5), skeletal l1 (600 -> 300, p .6), fusion (1600 -> 100, p .5) fwd + dW
    skipped = 2 * (2 * 1000 * 2000 * 0.5 + 2 * 600 * 1200 * 0.6 + 2 * 2 * 1600 * 400 * 0.5)
    skipped += 2 * (2 * 39 * 2000 * 0.4)      # audio l0: F = 39, p = .4 also takes the dropout-aware kernel
    skipped += 2 * (2 * 20 * 1200 * 0.6)      # skeletal l0: F = 20, p = .6
    assert abs((alg - ex) - skipped) < 1e-6 * alg


def test_bench_watchdog_ends_a_rank_that_stops_making_progress(tmp_path):
    """bench.py's Watchdog: beats keep the process alive; when they stop, the process exits with code 3 and says where it was."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Watchdog(0.4, 7)\n"
            "for i in range(6):\n"
            "    time.sleep(0.15); d.beat('step %%d' %% i)\n"      # 0.9 s of life with a 0.4 s limit: the beats are honoured
            "print('alive', flush=True)\n"
            "time.sleep(30)\n" % root)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert r.returncode == 3 and "alive" in r.stdout
    assert "rank 7 made no progress" in r.stderr and "last: step 5" in r.stderr
    assert time.time() - t0 < 10
    # limit 0 switches it off
    code0 = "import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(0, 0); time.sleep(0.5); print('ok')" % root
    r = subprocess.run([sys.executable, "-c", code0], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and "ok" in r.stdout


def test_bench_launcher_terminates_the_other_ranks_when_one_fails(tmp_path, monkeypatch):
    """bench._spawn_ranks: the first rank that exits non-zero (e.g. ended by its watchdog) takes the others with it - nobody is
    left waiting in a collective.  Ranks are fresh child processes of a launcher that never touched the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "open(os.path.join(%r, 'pid%%d' %% r), 'w').write(str(os.getpid()))\n"
                      "if r == 1:\n"
                      "    time.sleep(0.5); sys.exit(3)\n"
                      "time.sleep(60)\n" % str(tmp_path))
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "sys.argv = [%r]\n"
            "sys.exit(bench._spawn_ranks(3))\n" % (root, str(script)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=40)
    assert r.returncode == 3 and time.time() - t0 < 20
    time.sleep(0.3)
    for k in range(3):
        pid = int(open(tmp_path / ("pid%d" % k)).read())
        assert not os.path.exists("/proc/%d" % pid) or open("/proc/%d/stat" % pid).read().split()[2] == "Z", k


def test_bench_launcher_hands_out_distinct_local_ranks_and_rank_r_binds_device_r():
    """`python bench.py --gpus N` as its own launcher: N distinct RANK / LOCAL_RANK values, one rendezvous address for all; a
    rank binds the GPU of its LOCAL_RANK (one process per GPU) - more ranks than GPUs only with the host communicator, where
    they wrap around and `ranks_per_gpu` says so."""
    import bench
    for n in (1, 2, 4, 8):
        envs = bench.rank_environments(n, 29511, base={})
        assert [e["RANK"] for e in envs] == [str(r) for r in range(n)]
        assert [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(n)]
        assert {e["WORLD_SIZE"] for e in envs} == {str(n)} and {e["LOCAL_WORLD_SIZE"] for e in envs} == {str(n)}
        assert {(e["MASTER_ADDR"], e["MASTER_PORT"]) for e in envs} == {("127.0.0.1", "29511")}
        assert [bench.rank_device(int(e["LOCAL_RANK"]), n, n, "rccl") for e in envs] == [(r, 1) for r in range(n)]
    assert [bench.rank_device(r, 1, 2, "host") for r in range(2)] == [(0, 2), (0, 2)]
    assert [bench.rank_device(r, 2, 4, "host") for r in range(4)] == [(0, 2), (1, 2), (0, 2), (1, 2)]
    with pytest.raises(SystemExit):
        bench.rank_device(1, 1, 2, "rccl")          # RCCL refuses two ranks on one device: say so before it does
    with pytest.raises(SystemExit):
        bench.rank_device(0, 0, 1, "rccl")


def test_effective_cores_honours_the_cgroup_quota_and_import_caps_blas_pools(tmp_path):
    """mgr_amd/_hostenv.py: the GPU boxes show 256 cores and grant 16 cores' worth of CPU time; a BLAS pool sized by the core
    count got the whole process frozen for tens of ms (profiles/r03_host_stalls.txt).  Importing the package (before numpy
    starts its pools) caps OPENBLAS / OMP / MKL_NUM_THREADS at the quota unless the caller has chosen a size."""
    import subprocess
    import sys as _sys
    from mgr_amd import _hostenv
    n = _hostenv.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert n <= max(1, int(quota) // int(period))
    except OSError:
        pass
    code = ("import os, sys; sys.path.insert(0, %r); import mgr_amd; "
            "print(os.environ['OPENBLAS_NUM_THREADS'], os.environ['OMP_NUM_THREADS'])" % ROOT)
    env = {k: v for k, v in os.environ.items()
           if k not in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS", "LOCAL_WORLD_SIZE", "WORLD_SIZE")}
    out = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == [str(n), str(n)]
    env["OMP_NUM_THREADS"] = "3"       # a caller's own choice stays
    out = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == [str(n), "3"]
    del env["OMP_NUM_THREADS"]
    env["LOCAL_WORLD_SIZE"] = "4"      # the ranks of a node share the quota
    out = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == [str(max(1, n // 4))] * 2


def test_rank_aware_generator_assembles_only_its_shard_of_every_global_batch():
    """DataGenerator(..., rank=, world=): every rank walks the SAME file lists (seeded shuffle, split, whole-minibatch rule,
    wrap-around, epoch-end reshuffle) and assembles only its contiguous slice of each global minibatch:
    rank-aware batch == parallel.shard_batch(full batch) - for training batches, validation batches and a short last batch of
    an un-truncated list (reference per-file loop: multimodal_fusion/data_generator.py:157-278)."""
    import random
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    from mgr_amd.parallel import shard_batch
    world, mb = 2, 8
    kw = dict(synthetic_files=45)
    random.seed(123)
    full = DataGenerator(mb, 20, 39, 40, 22, 'train', **kw)
    parts = [DataGenerator(mb, 20, 39, 40, 22, 'train', rank=r, world=world, **kw) for r in range(world)]
    for g in parts:
        assert g.get_size(True) == full.get_size(True) and g.get_file_list(True) == full.get_file_list(True)
        assert g.get_file_list(False) == full.get_file_list(False)

    def same(a, b):
        assert set(a[0]) == set(b[0])
        for k in a[0]:
            assert a[0][k].shape == b[0][k].shape and np.array_equal(a[0][k], b[0][k]), k
        assert a[1]['ctc'].shape == b[1]['ctc'].shape

    for train in (True, False):
        gens = [(g.next_train() if train else g.next_val()) for g in [full] + parts]
        for _ in range(full.get_size(train) // mb + 2):          # (+2: across the wrap-around)
            fb = next(gens[0])
            fb = ({k: v.copy() for k, v in fb[0].items()}, fb[1])
            for r in range(world):
                rb = next(gens[1 + r])
                assert rb[0]['the_input_audio'].shape[0] == mb // world
                same(rb, (shard_batch(fb[0], r, world), {'ctc': np.zeros(mb // world)}))
    # epoch end: the same global `random` state gives every rank the same new order
    lists = []
    for g in [full] + parts:
        random.seed(5)
        g.on_epoch_end(0)
        lists.append((list(g.train_list), list(g.val_list)))
    assert lists[0] == lists[1] == lists[2]
    # a short last batch of an un-truncated list ('final': 14 files, global batch 8 -> 8, then 6 -> 3 + 3)
    vf = DataGenerator(mb, 20, 39, 40, 22, 'final', synthetic_files=14)
    vp = [DataGenerator(mb, 20, 39, 40, 22, 'final', synthetic_files=14, rank=r, world=world) for r in range(world)]
    for g in [vf] + vp:
        g.val_index = 8
    fb = vf.get_batch(False)
    assert fb[0]['the_labels'].shape[0] == 6
    for r in range(world):
        same(vp[r].get_batch(False), (shard_batch(fb[0], r, world), {'ctc': np.zeros(3)}))
    with pytest.raises(ValueError):
        DataGenerator(9, 20, 39, 40, 22, 'train', rank=0, world=2, **kw)      # global batch not divisible
    with pytest.raises(ValueError):
        DataGenerator(8, 20, 39, 40, 22, 'train', rank=2, world=2, **kw)


def test_training_after_compile_with_rmsprop_is_refused():
    """The decode scripts compile with RMSprop and never train (sequence_decoding.py:112-115); fit_generator / train_on_batch on
    such a model must refuse instead of silently running the device's Adam with default settings."""
    from mgr_amd.keras_like import RMSprop
    spec = configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=4)
    m = Model(spec)
    m.compile(loss={'ctc': lambda a, b: b}, optimizer=RMSprop(lr=0.01))
    with pytest.raises(NotImplementedError) as ei:
        m.fit_generator(iter(()), steps_per_epoch=1, epochs=1)
    assert "RMSprop" in str(ei.value) and "Adam" in str(ei.value)
    with pytest.raises(NotImplementedError):
        m.train_on_batch({"the_labels": np.zeros((1, 3))})
    m.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-4))       # compiling with Adam afterwards is fine
    m._require_trainable()


def test_only_rank_zero_writes_checkpoints(tmp_path, monkeypatch):
    """Data parallel: every replica takes the same save_best_only decision (the validation loss is all-reduced), rank 0 alone
    writes - Model.save_weights and the generator's epoch-end model files (multimodal.py:252-258, data_generator.py:317-321)."""
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator

    class Comm:
        def __init__(self, rank):
            self.rank, self.dev = rank, None

    spec = configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=4)
    monkeypatch.chdir(tmp_path)
    for rank in (1, 0):
        m = Model(spec)
        m.distribute(Comm(rank), 2)
        assert m.is_chief == (rank == 0)
        ck = ModelCheckpoint(str(tmp_path / "best.h5"), monitor='val_loss', save_best_only=True, save_weights_only=True)
        ck.set_model(m)
        ck.on_epoch_end(0, {"val_loss": 2.0})
        assert ck.best == 2.0                                           # the decision is taken on every rank ...
        assert os.path.exists(tmp_path / "best.h5") == (rank == 0)      # ... the file is written by rank 0
        g = DataGenerator(4, 20, 39, 30, 22, 'train', synthetic_files=12, rank=rank, world=2)
        g.model = m
        g.on_epoch_end(0)
        assert os.path.exists(tmp_path / g.model_json_name) == (rank == 0)


def test_local_world_size_does_not_take_a_multi_node_world_for_the_node(monkeypatch):
    from mgr_amd import _hostenv
    for k in ("LOCAL_WORLD_SIZE", "WORLD_SIZE", "NNODES", "GROUP_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert _hostenv.local_world_size() == 1
    monkeypatch.setenv("WORLD_SIZE", "16")                 # 2 x 8 ranks, no LOCAL_WORLD_SIZE: not this node's rank count
    assert _hostenv.local_world_size() == 1
    monkeypatch.setenv("NNODES", "1")
    assert _hostenv.local_world_size() == 16
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _hostenv.local_world_size() == 8
    # opt-out of the import side effect
    import subprocess
    import sys as _sys
    code = "import os, sys; sys.path.insert(0, %r); import mgr_amd; print(os.environ.get('OPENBLAS_NUM_THREADS'))" % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    env["MGR_NO_THREAD_CAP"] = "1"
    assert subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split() == ["None"]


def test_binding_constants_and_structs_match_the_header():
    """The ctypes binding restates enums and struct layouts of include/mgr.h by hand: the launch forms, MGR_SEQ_NONE, the ABI revision,
    the profiling families, and the member lists of the three structs it fills field by field (the library reports their sizes at
    load time - mgr_abi_struct_sizes - but only a GPU box loads it; this is the CPU-side half of the guard)."""
    import re
    from mgr_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "mgr.h")).read()
    flat = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)

    def enum(name):
        m = re.search(r"\b%s\s*=\s*(0x[0-9A-Fa-f]+|\d+)" % name, flat)
        assert m, name
        return int(m.group(1), 0)

    for c_name, py in (("MGR_SCAN_FORM_AUTO", _capi.SCAN_FORM_AUTO), ("MGR_SCAN_FORM_PLAIN", _capi.SCAN_FORM_PLAIN),
                       ("MGR_SCAN_FORM_PAIR", _capi.SCAN_FORM_PAIR), ("MGR_SCAN_FORM_FUSED", _capi.SCAN_FORM_FUSED),
                       ("MGR_SCAN_FORM_FUSED_ANY", _capi.SCAN_FORM_FUSED_ANY), ("MGR_BPTT_FORM_AUTO", _capi.BPTT_FORM_AUTO),
                       ("MGR_BPTT_FORM_TRIMMED", _capi.BPTT_FORM_TRIMMED), ("MGR_BPTT_FORM_YIELDING", _capi.BPTT_FORM_YIELDING),
                       ("MGR_BPTT_FORM_DIRECT", _capi.BPTT_FORM_DIRECT), ("MGR_BPTT_FORM_FUSED", _capi.BPTT_FORM_FUSED),
                       ("MGR_BPTT_FORM_FUSED_DIRECT", _capi.BPTT_FORM_FUSED_DIRECT), ("MGR_BPTT_FORM_SINGLE_CU", _capi.BPTT_FORM_SINGLE_CU),
                       ("MGR_SCAN_GAVE_UP", _capi.SCAN_GAVE_UP), ("MGR_SCAN_NONFINITE", _capi.SCAN_NONFINITE),
                       ("MGR_K_SCAN_FWD", _capi.K_SCAN_FWD), ("MGR_K_SCAN_FWD_NARROW", _capi.K_SCAN_FWD_NARROW),
                       ("MGR_K_ALLREDUCE", _capi.K_ALLREDUCE)):
        assert enum(c_name) == py, c_name
    assert enum("MGR_K_COUNT") == len(_capi.KERNEL_FAMILIES)
    assert int(re.search(r"#define MGR_ABI_REVISION\s+(\d+)", hdr).group(1)) == _capi.ABI_REVISION
    assert int(re.search(r"#define MGR_SEQ_NONE\s+(0x[0-9A-Fa-f]+)u", hdr).group(1), 16) == _capi.SEQ_NONE

    def members(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), flat, re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = [p.strip() for p in decl.split(",")]
            names.append(re.search(r"(\w+)$", first.replace("*", " ")).group(1))
            names += [re.search(r"(\w+)$", r.replace("*", " ")).group(1) for r in rest]
        return names

    assert members("mgr_scan_job") == [n for n, _ in _capi.ScanJob._fields_]
    assert members("mgr_scan_bwd_job") == [n for n, _ in _capi.ScanBwdJob._fields_]
    assert members("mgr_scan_launch_opts") == [n for n, _ in _capi.ScanLaunchOpts._fields_]
    o = _capi.make_launch_opts(_capi.SCAN_FORM_FUSED, 0)
    assert o.struct_size == __import__("ctypes").sizeof(_capi.ScanLaunchOpts) and o.form == 3 and not o.seq_out
