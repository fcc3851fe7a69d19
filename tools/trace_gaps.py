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

# --context N: for every gap longer than 100 us, the N kernels before and after it (name, queue, duration): what the GPU waited for
if "--context" in sys.argv:
    n_ctx = int(sys.argv[sys.argv.index("--context") + 1])
    full = []
    for r in csv.DictReader(open(sys.argv[1])):
        full.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70],
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    full.sort()
    full = full[int(len(full) * skip):]
    end = full[0][1]
    shown = 0
    for i in range(1, len(full)):
        s, e, n, q, st = full[i]
        if s > end and s - end > 100000 and shown < 12:
            shown += 1
            print(f"--- gap of {(s - end) / 1e3:.1f} us")
            for j in range(max(0, i - n_ctx), min(len(full), i + n_ctx)):
                mark = ">>" if j == i else "  "
                print(f"   {mark} start +{(full[j][0] - full[i][0]) / 1e3:9.1f} us  dur {(full[j][1] - full[j][0]) / 1e3:8.1f} us  queue {full[j][3]} stream {full[j][4]}  {full[j][2]}")
        end = max(end, e)
