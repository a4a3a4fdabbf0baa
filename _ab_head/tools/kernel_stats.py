"""python tools/kernel_stats.py <rocprofv3 *_kernel_stats.csv> [name fragment ...]: calls, avg / min / max ms per kernel (all, or those
whose name contains one of the fragments).  min is close to the kernel's time alone on the chip when a trace holds an un-overlapped step."""
import csv
import sys
frags = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if not frags or any(f in n for f in frags):
        print(n[:56].ljust(56), "%5s" % r["Calls"], "avg %8.3f  min %8.3f  max %8.3f ms  %5.1f %%" % (
            float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, float(r["Percentage"])))
