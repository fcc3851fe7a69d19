"""Aggregate a rocprofv3 kernel_stats.csv by kernel family, so the profile can be compared with bench.py's roofline
object (whose `avg_launch_ms` is the mean over every launch of the GEMM family).  Usage:
    python tools/profile_family.py profiles/r01_final_bench_small_kernel_stats.csv"""
import csv
import sys

FAMILIES = [("gemm", ("gemm_v2_kernel", "gemm_v3_", "gemm_v4_kernel", "gemm_v7_kernel", "gemm_bf16_kernel")), ("splitk_reduce", ("splitk_reduce",)),
            ("attn_fwd", ("attn_fwd_kernel",)),
            ("attn_bwd", ("attn_bwd_", "attn_delta", "attn_dq_reduce")),   # (attn_bwd_prep_kernel, attn_bwd_fused_kernel, the dQ / dK/dV pair)
            ("layernorm", ("ln_",)),
            ("masked_ce", ("masked_ce",)), ("adamw/sumsq", ("adamw", "sumsq")), ("embedding", ("embed",))]


def main(path):
    tot = {}
    allns = 0
    with open(path) as f:
        for r in csv.DictReader(f):
            name, calls, ns = r["Name"], int(r["Calls"]), int(r["TotalDurationNs"])
            allns += ns
            fam = next((n for n, keys in FAMILIES if any(k in name for k in keys)), "other (torch glue)")
            c, t = tot.get(fam, (0, 0))
            tot[fam] = (c + calls, t + ns)
    print(f"{'family':22s} {'calls':>8s} {'total ms':>10s} {'avg us':>9s} {'share':>7s}")
    for fam, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{fam:22s} {c:8d} {t / 1e6:10.2f} {t / c / 1e3:9.1f} {100.0 * t / allns:6.1f}%")


if __name__ == "__main__":
    main(sys.argv[1])
