"""Median / min duration per kernel name from a rocprofv3 --kernel-trace CSV directory.   python tools/ktrace_summary.py DIR [substring ...]"""
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    v.sort()
    print(f"{k[:90]:90s} n {len(v):5d}  median {v[len(v) // 2]:8.2f} us  min {v[0]:8.2f}  p90 {v[int(len(v) * 0.9)]:8.2f}")
