#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_* (bench JSON, rocprofv3 kernel stats, PMC passes) into profiles/<tag>_summary.md
and copy the raw per-kernel CSV summaries next to it."""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
G = "gpurun_out"
os.makedirs("profiles", exist_ok=True)


def short(name):
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[A-Za-z]+)", name)
    return m.group(1) if m else name[:40]


out = ["# %s — bench + rocprofv3 summary (MI355X, config F: B=64, T=1900)\n" % tag]
j = json.loads(open("%s/%s_bench.json" % (G, tag)).read())
out.append("## bench.py line\n```json\n%s\n```\n" % json.dumps(j, indent=1))
shutil.copy("%s/%s_kernel_stats.csv" % (G, tag), "profiles/%s_kernel_stats.csv" % tag)
rows = list(csv.DictReader(open("%s/%s_kernel_stats.csv" % (G, tag))))
out.append("## rocprofv3 --kernel-trace --stats (the default bench.py run: 50 timed steps + 3 warm-up)\n")
out.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:14]:
    out.append("| %s | %s | %.2f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))


def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    path = "%s/%s_pmc_%s.csv" % (G, tag, name)
    if not os.path.exists(path):
        return agg, cnt
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_scan_cluster_k16f"):
            # (the fused scan kernel runs the encoder depths - 208 workgroups - AND, since round 6, the fusion layer's own scan - 32: two
            #  different launches under one name; a per-launch average over both would describe neither)
            k += "@%d" % (int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
    return agg, cnt


f, fc = pmc("FETCH_SIZE")
w, wc = pmc("WRITE_SIZE")
out.append("\n## HBM traffic per launch (PMC, separate passes of bench.py --steps 8 --warmup 3 - the fused schedule engages from the second step on; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)\n")
out.append("| kernel | launches | fetch MB/launch (x2-corrected) | write MB/launch |\n|---|---|---|---|")
for k in sorted(f, key=lambda k: -f[k]["FETCH_SIZE"]):
    if fc[k] == 0:
        continue
    out.append("| %s | %d | %.1f | %.1f |" % (k, fc[k], 2 * f[k]["FETCH_SIZE"] / fc[k] / 1024,
                                           w[k]["WRITE_SIZE"] / max(1, wc[k]) / 1024))
# per-family HBM bytes per launch for bench.py's roofline.traffic (read + write, FETCH_SIZE x2-corrected, KB -> bytes): the figure
# of the family's DOMINANT KERNEL in the full run's kernel trace - never a mix of forms (round 5's passes were too short for the fused
# schedule to engage and described a kernel that was no longer on the hot path: VERDICT r05)
def family_of(k):   # (kernel names carry variant suffixes: _ks, _k16, _k16f, _k16fs, _s, 16_split, ...)
    if k.startswith("k_scan_cluster_bwd") or k.startswith("k_scan_bwd"):
        return "scan_bwd"
    if k.startswith("k_scan_cluster") or k.startswith("k_scan_"):
        if "@" in k:
            return "scan_fwd" if int(k.split("@")[1]) > 64 else "scan_fwd_narrow"
        return "scan_fwd_narrow" if (k.endswith("_s") or k.endswith("k16_s")) else "scan_fwd"
    for prefix, fam in (("k_gemm_nn", "gemm_nn"), ("k_gemm_tn", "gemm_tn"), ("k_proj_split", "gemm_nn"), ("k_dw_split", "gemm_tn"),
                        ("k_gemm_nt", "gemm_nt")):
        if k.startswith(prefix):
            return fam
    return None


run_ns = collections.defaultdict(float)         # total time per (short) kernel name in the FULL run (rocprofv3 --stats)
for r in rows:
    run_ns[short(r["Name"])] += float(r["TotalDurationNs"])
dom_of = {}                                     # family -> its kernel with the largest total time in the full run
for k, ns in run_ns.items():
    fam = family_of(k)
    if fam and ns > run_ns.get(dom_of.get(fam), 0.0):
        dom_of[fam] = k
kernel_bytes = {k: (2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0 / fc[k] for k in f if fc[k]}
# a family's dominant kernel of the full run (names without a grid) -> its PMC entry: the plain name, or the name @ its largest grid
pmc_of = {}
for fam, k in list(dom_of.items()):
    cands = [q for q in kernel_bytes if q == k or q.startswith(k + "@")]
    cands = [q for q in cands if family_of(q) == fam] or cands
    if cands:
        pmc_of[fam] = max(cands, key=lambda q: int(q.split("@")[1]) if "@" in q else 0)
if kernel_bytes:
    import subprocess
    sys.path.insert(0, ".")
    import mgr_amd  # noqa: F401
    from mgr_amd._build import source_hash
    dominant = max((k for k in run_ns if family_of(k)), key=lambda k: run_ns[k])
    if not any(q == dominant or q.startswith(dominant + "@") for q in kernel_bytes):
        sys.exit("summarize_profile: the PMC passes hold no dispatch of %s, the dominant kernel of the full run (they hold %s): "
                 "NOT writing profiles/pmc_traffic.json - take the passes on the schedule the product runs" % (dominant, sorted(kernel_bytes)))
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        head = None
    # src_sha ties the numbers to the tree they were measured on (written on the GPU box by profile_round.sh): bench.py reports
    # `traffic` only when its own tree hashes the same
    sha_file = "%s/%s_src_sha.txt" % (G, tag)
    src_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else source_hash()
    json.dump({"tag": tag, "src_sha": src_sha, "head_at_summary": head,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 8 --warmup 3; per family: the family's dominant kernel of the full run",
               "dominant_kernel": dominant, "family_kernel": dict(pmc_of),
               "pmc_dispatches": {k: fc[k] for k in kernel_bytes if family_of(k)},
               "bytes_per_launch": {fam: kernel_bytes[q] for fam, q in pmc_of.items()},
               "kernel_bytes_per_launch": {k: v for k, v in kernel_bytes.items() if family_of(k)}},
              open("profiles/pmc_traffic.json", "w"), indent=1)
    out.append("\nDominant kernel of the full run: `%s`; PMC entry per family (name @ workgroups where one kernel runs two kinds of launch): %s\n" % (
        dominant, ", ".join("%s = %s (%d dispatches)" % (fam, q, fc[q]) for fam, q in sorted(pmc_of.items()))))
s, sc = pmc("SQ_VALU_MFMA_BUSY_CYCLES")
out.append("\n## SQ counters per kernel (sums over launches)\n")
out.append("| kernel | launches | MFMA busy / (1024 SIMD x GUI_ACTIVE/8) | WAIT_INST_ANY/WAVE_CYCLES | WAIT_ANY/WAVE_CYCLES | ACTIVE/WAVE_CYCLES | LDS bank conflict cycles |\n|---|---|---|---|---|---|---|")
for k in sorted(s, key=lambda k: -s[k]["SQ_VALU_MFMA_BUSY_CYCLES"]):
    v = s[k]
    if v["GRBM_GUI_ACTIVE"] == 0 or v["SQ_WAVE_CYCLES"] == 0:
        continue
    out.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.3g |" % (
        k, sc[k], v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8), v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
        v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_LDS_BANK_CONFLICT"]))
# ---- the other BASELINE configurations, decode, fit_generator, 2-rank host run (written by profile_round.sh without "quick")
import glob
extra = []
for f in sorted(glob.glob("%s/%s_cfg_*_bench.json" % (G, tag))):
    cfg = os.path.basename(f)[len(tag) + 5:-len("_bench.json")]    # ("E", "S_ref", ... and "E_b64", ...: the same network at B = 64)
    try:
        d = json.loads(open(f).read())
    except ValueError:
        continue
    shutil.copy(f, "profiles/%s_cfg_%s_bench.json" % (tag, cfg))
    ks = "%s/%s_cfg_%s_kernel_stats.csv" % (G, tag, cfg)
    top = ""
    if os.path.exists(ks):
        shutil.copy(ks, "profiles/%s_cfg_%s_kernel_stats.csv" % (tag, cfg))
        rr = list(csv.DictReader(open(ks)))[:3]
        top = "; ".join("%s %.2f ms avg x %s" % (short(r["Name"]), float(r["AverageNs"]) / 1e6, r["Calls"]) for r in rr)
    roof = d.get("roofline") or {}
    extra.append("| %s (B = %s) | %.3f | %.0f | %s %.3f | %s | %s |" % (cfg, d["config"].get("per_gpu_batch"), d["ms_per_step"], d["value"], roof.get("kernel"), roof.get("frac") or 0.0,
                                                       (d.get("ctc_loss_parity") or {}).get("rel_delta"), top))
if extra:
    out.append("\n## Other BASELINE configurations (bench.py --config, 10 steps; parity cases, not the headline)\n")
    out.append("| config | ms/step | frames/s | dominant family, frac of f32-MFMA peak | CTC loss rel. delta vs fp64 oracle | top kernels (rocprofv3 --stats, 5 steps) |\n|---|---|---|---|---|---|")
    out += extra
for name in ("decode.json", "decode_kernel_stats.csv", "fit.txt", "dp2_host.json"):
    f = "%s/%s_%s" % (G, tag, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, "profiles/%s_%s" % (tag, name))
open("profiles/%s_summary.md" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out[3:]))
