"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (same number of steps): where does variant B spend more?
    python tools/kstats_diff.py A_kernel_stats.csv B_kernel_stats.csv [steps]"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        n = re.sub(r"\(.*", "", n)
        n = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)
        c, t = d.get(n, (0, 0)); d[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    rows.append(((tb - ta) / 1e6 / steps, k, ca, ta / 1e6 / steps, cb, tb / 1e6 / steps))
rows.sort(reverse=True)
print(f"{'kernel':60s} {'calls A':>8s} {'ms A':>9s} {'calls B':>8s} {'ms B':>9s} {'B - A ms':>9s}")
for d, k, ca, ta, cb, tb in rows:
    if abs(d) >= 0.02:
        print(f"{k[:60]:60s} {ca:8d} {ta:9.2f} {cb:8d} {tb:9.2f} {d:9.2f}")
print(f"total A {sum(v[1] for v in a.values()) / 1e6 / steps:.2f} ms, B {sum(v[1] for v in b.values()) / 1e6 / steps:.2f} ms per step")
