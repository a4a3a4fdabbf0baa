"""Timeline of ONE steady-state training step from a rocprofv3 kernel trace (tools/ab_bench.sh writes gpurun_out/<tag>_v<i>_kernel_trace.csv):
    python tools/step_timeline.py gpurun_out/tl_v0_kernel_trace.csv [step_index_from_end]
Kernels are listed per hardware queue in start order with start offset and duration (ms); kernels shorter than 30 us are folded into
a count.  The step window runs from one Adam launch (k_adam) to the next."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "k_adam" in r["Kernel_Name"]]
# the LAST k_adam launch of every step: launches closer than 1 ms belong to one step
ends = [a for i, a in enumerate(adam) if i + 1 == len(adam) or adam[i + 1]["s"] - a["s"] > 2_000_000]
w0, w1 = ends[-back - 1]["e"], ends[-back]["e"]
print("step window %.3f ms" % ((w1 - w0) / 1e6))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]
qs = {}
for r in rows:
    if r["e"] <= w0 or r["s"] >= w1:
        continue
    qs.setdefault(r["Queue_Id"], []).append(r)
for q, lst in sorted(qs.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    busy = sum(min(r["e"], w1) - max(r["s"], w0) for r in lst) / 1e6
    print("queue %s: %d kernels, busy %.2f ms" % (q, len(lst), busy))
    small = 0
    for r in lst:
        d = (r["e"] - r["s"]) / 1e6
        if d < 0.03:
            small += 1
            continue
        print("   +%7.3f  %7.3f ms  %-44s grid %s" % ((r["s"] - w0) / 1e6, d, short(r["Kernel_Name"]), r["Grid_Size_X"]))
    if small:
        print("   (%d kernels under 30 us)" % small)
