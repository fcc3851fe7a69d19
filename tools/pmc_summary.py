"""Summarise rocprofv3 --pmc passes: per kernel name, the mean of every counter over its dispatches (skipping the first
`--skip` dispatches of each kernel: warm-up), plus a few ratios the MI355X guide reads them by.
    python tools/pmc_summary.py [--skip 2] [--match attn] pass1_counter_collection.csv [pass2_counter_collection.csv ...]
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(= 32 x N_mfma for v_mfma_f32_32x32x16_bf16) summed over SIMDs (MI355X_MICROARCH.md, constants table)."""
import collections
import csv
import sys


def main(argv):
    skip, match = 2, ""
    while argv and argv[0] in ("--skip", "--match"):
        if argv[0] == "--skip":
            skip = int(argv[1])
        else:
            match = argv[1]
        argv = argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [per dispatch]
    for path in argv:
        per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> value
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if match and match not in name:
                continue
            per[(name, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
        seen = collections.Counter()
        for (name, did), cs in sorted(per.items(), key=lambda kv: kv[0][1]):
            seen[name] += 1
            if seen[name] <= skip:
                continue
            for c, v in cs.items():
                vals[name][c].append(v)
    for name, cs in sorted(vals.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        n = len(next(iter(cs.values())))
        print(f"== {name}   ({n} dispatches averaged)")
        for c in sorted(m):
            print(f"   {c:34s} {m[c]:16.0f}")
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if c in m:
                    print(f"   {c + ' / SQ_WAVE_CYCLES':34s} {m[c] / wc:16.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            print(f"   {'MFMA busy / SQ_BUSY_CYCLES':34s} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:16.3f}   (per-SE busy cycles as the denominator: compare between builds, not as an absolute)")
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   {'LDS bank-conflict share':34s} {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:16.3f}")


if __name__ == "__main__":
    main(sys.argv[1:])
