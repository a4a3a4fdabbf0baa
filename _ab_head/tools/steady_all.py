"""Every kernel (also the short ones) of ONE steady-state step on one hardware queue of a rocprofv3 kernel trace, with the gaps between them:
python tools/steady_all.py <trace.csv> <queue id> [from_ms to_ms] [steps_from_end=5]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
q = sys.argv[2]
lo, hi = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.0, 1e9)
back = int(sys.argv[5]) if len(sys.argv) > 5 else 5
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "k_adam" in r["Kernel_Name"]]
ends = [a for i, a in enumerate(adam) if i + 1 == len(adam) or adam[i + 1]["s"] - a["s"] > 2_000_000]
w0, w1 = ends[-back]["e"], ends[-back + 1]["e"]
prev = None
tot = gap = 0.0
for r in rows:
    if r["Queue_Id"] != q or r["e"] <= w0 or r["s"] >= w1:
        continue
    off = (r["s"] - w0) / 1e6
    if not (lo <= off <= hi):
        continue
    d = (r["e"] - r["s"]) / 1e6
    g = (r["s"] - prev) / 1e6 if prev else 0.0
    prev = r["e"]
    tot += d
    gap += max(g, 0.0)
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
    print("+%7.3f  %6.3f  (gap %6.3f)  %s" % (off, d, g, re.sub(r"\(.*", "", n)[:60]))
print("kernels %.3f ms, gaps %.3f ms" % (tot, gap))
