"""GPU idle time inside a rocprofv3 kernel trace: union of the kernel intervals against the span of the trace, the largest
gaps, and which kernels follow them.    python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]          # the last part of the trace: steady-state steps
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy (union) {busy / 1e6:.2f} ms  idle {100.0 * (span - busy) / span:.2f} %")
tot = {}
for g, n in gaps:
    t = tot.setdefault(n, [0, 0]); t[0] += g; t[1] += 1
print("idle time by the kernel that ends the gap (top 12):")
for n, (g, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {g / 1e6:8.3f} ms in {c:5d} gaps (avg {g / c / 1e3:6.1f} us)  before {n}")
