"""Long kernels (>= 0.15 ms) of ONE steady-state training step of a rocprofv3 kernel trace (tools/trace_steps.sh), by hardware queue:
    python tools/steady_timeline.py gpurun_out/<tag>_kernel_trace.csv [steps_from_end=5]
The window runs from the end of one step's last k_adam launch to the next one's; steps near the end of a run are not steady state
(the last timed step does not prefetch), hence the default of five steps back."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "k_adam" in r["Kernel_Name"]]
ends = [a for i, a in enumerate(adam) if i + 1 == len(adam) or adam[i + 1]["s"] - a["s"] > 2_000_000]
print("steps (ms):", [round((b["e"] - a["e"]) / 1e6, 2) for a, b in zip(ends, ends[1:])])
w0, w1 = ends[-back]["e"], ends[-back + 1]["e"]
print("window %.3f ms" % ((w1 - w0) / 1e6))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:40]


busy = {}
for r in rows:
    if r["e"] <= w0 or r["s"] >= w1:
        continue
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + (min(r["e"], w1) - max(r["s"], w0)) / 1e6
    d = (r["e"] - r["s"]) / 1e6
    if d >= 0.15:
        print("q%-3s +%8.3f %7.3f  %s" % (r["Queue_Id"], (r["s"] - w0) / 1e6, d, short(r["Kernel_Name"])))
print("busy per queue (ms):", {k: round(v, 2) for k, v in busy.items()})
