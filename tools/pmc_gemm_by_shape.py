"""HBM-side traffic of the step's GEMM launches, attributed to SHAPES: which launches of the GEMM family read
more than their operands, and by how much.

Two modes:

  run     python tools/pmc_gemm_by_shape.py run [--rows 32768] [--plans profiles/r05_gemm_plans_small.json]
          launches every distinct GEMM of one 32-row pass of the small config once per REPS, in a fixed order, under whatever
          profiler wraps this process.  Use it as the command of two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a
          pass: TCC slots), e.g.
              rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -o x -- python3 tools/pmc_gemm_by_shape.py run
              rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -o x -- python3 tools/pmc_gemm_by_shape.py run
  parse   python tools/pmc_gemm_by_shape.py parse <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
          maps the GEMM dispatches of the two passes, in dispatch order, back to the shapes (every shape contributes the same
          number of kernel dispatches in both passes) and writes, per shape: read bytes (2 x FETCH_SIZE: gfx950 tallies wide
          coalesced reads at half their bytes, MI355X_MICROARCH.md "HBM"), written bytes (WRITE_SIZE, exact), the algorithmic
          bytes (operands + outputs + epilogue operands, each once) and the ratios.

FETCH_SIZE counts fabric-side requests of the XCDs' L2s (Infinity-Cache hits included): it is what leaves the L2s, not what
reaches HBM.  Launches run back to back here, so an operand a previous launch left in the 256-MiB Infinity Cache still counts."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 3
C, V = 1024, 65536


def shapes(rows):
    """(name, kind, M, N, K, a_kmajor, b_kmajor, epilogue name, operand bytes read once + output bytes written once)"""
    M = rows
    Mm = max(64, int(round(0.15 * rows / 8)) * 8)
    b2 = 2
    out = []

    def add(name, m, n, k, ak, bk, epi, aux_reads=0, outs=1):
        alg = (m * k + n * k) * b2 + outs * m * n * b2 + aux_reads * m * n * b2
        out.append(dict(name=name, M=m, N=n, K=k, a_kmajor=ak, b_kmajor=bk, epi=epi, algorithmic_bytes=alg,
                        algorithmic_read_bytes=(m * k + n * k + aux_reads * m * n) * b2, algorithmic_write_bytes=outs * m * n * b2))
    add("c_attn + RoPE (fwd)", M, 3 * C, C, True, True, "rope")
    add("attn c_proj + residual (fwd)", M, C, C, True, True, "add", aux_reads=1)
    add("c_fc + GELU (fwd)", M, 4 * C, C, True, True, "gelu", outs=2)
    add("mlp c_proj + residual (fwd)", M, C, 4 * C, True, True, "add", aux_reads=1)
    add("mlp c_proj dgrad + GELU' (bwd)", M, 4 * C, C, True, False, "gelu_bwd", aux_reads=1)
    add("c_fc dgrad (bwd)", M, C, 4 * C, True, False, "none")
    add("attn c_proj dgrad (bwd)", M, C, C, True, False, "none")
    add("c_attn dgrad (bwd)", M, C, 3 * C, True, False, "none")
    add("readout fwd, masked rows", Mm, V, C, True, True, "none")
    add("readout dgrad, masked rows", Mm, C, V, True, False, "none")
    return out


def run(rows, plans):
    import torch
    from omnibiote_amd import _lib as L, ops, tune
    if plans and os.path.exists(plans):
        tune.load_plans(plans)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    epi_of = {"none": L.EPI_NONE, "add": L.EPI_ADD, "gelu": L.EPI_GELU, "gelu_bwd": L.EPI_GELU_BWD, "rope": L.EPI_ROPE_QK}
    tab = torch.randn(1024, 64, device=dev, generator=g)
    rope = (torch.cos(tab), torch.sin(tab), 1024, 128)
    for sh in shapes(rows):
        M, N, K = sh["M"], sh["N"], sh["K"]
        a = torch.randn(M * K, device=dev, generator=g).to(torch.bfloat16)
        b = torch.randn(N * K, device=dev, generator=g).to(torch.bfloat16)
        aux = torch.randn(M * N, device=dev, generator=g).to(torch.bfloat16) if sh["epi"] in ("add", "gelu_bwd") else None
        out = torch.empty(M * N, device=dev, dtype=torch.bfloat16)
        for _ in range(REPS):
            ops.gemm(a, b, M, N, K, sh["a_kmajor"], sh["b_kmajor"], epi_of[sh["epi"]], aux, out=out, rope=rope if sh["epi"] == "rope" else None)
        torch.cuda.synchronize()
        del a, b, aux, out
    # the block backward's grouped launch (four weight gradients + the c_attn input gradient), as the block issues it
    K = rows
    shp = [(4 * C, C), (C, 4 * C), (3 * C, C), (C, C)]
    data = [((torch.randn(K, m, device=dev, generator=g) * 0.5).to(torch.bfloat16), (torch.randn(K, n, device=dev, generator=g) * 0.5).to(torch.bfloat16),
             torch.zeros(m, n, device=dev, dtype=torch.bfloat16)) for m, n in shp]
    probs = [dict(a=a, b=b, M=m, N=n, K=K, out=o, accumulate=True) for (m, n), (a, b, o) in zip(shp, data)]
    for _ in range(REPS):
        ops.gemm_grouped(probs)
    torch.cuda.synchronize()


GEMM = ("gemm_v2_kernel", "gemm_v3_", "gemm_v4_kernel", "gemm_v7_kernel", "gemm_bf16_kernel", "splitk_reduce")


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if any(x in r["Kernel_Name"] for x in GEMM):
            rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def parse(fetch_csv, write_csv, out, rows=32768):
    f, w = load(fetch_csv), load(write_csv)
    assert len(f) == len(w), (len(f), len(w))
    sh = shapes(rows)
    C4 = [(4 * C, C), (C, 4 * C), (3 * C, C), (C, C)]
    grp_alg_r = sum((m + n) * rows * 2 + m * n * 2 for m, n in C4)     # operands once + the gradient accumulated into
    grp_alg_w = sum(m * n * 2 for m, n in C4)
    sh.append(dict(name="block weight gradients, grouped (bwd)", M=sum(m for m, n in C4), N=C, K=rows, epi="accumulate",
                   algorithmic_read_bytes=grp_alg_r, algorithmic_write_bytes=grp_alg_w, algorithmic_bytes=grp_alg_r + grp_alg_w))
    # dispatches per shape: equal consecutive groups (split-K shapes add their reduce kernel: still REPS x a constant)
    per = len(f) // len(sh) if len(f) % len(sh) == 0 else None
    res = {"correction": "read = 2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM), write = WRITE_SIZE; separate --pmc passes; KiB -> bytes",
           "rows_per_launch": rows, "shapes": []}
    i = 0
    for s in sh:
        # consume dispatches until the kernel name changes family or REPS launches were seen (split-K: launch + reduce)
        names = []
        j = i
        if per is not None:
            j = i + per
        else:
            first = f[i][1]
            seen = 0
            while j < len(f) and seen < REPS:
                if f[j][1] == first:
                    seen += 1
                j += 1
            while j < len(f) and "splitk_reduce" in f[j][1]:
                j += 1
        fr = sum(x[2] for x in f[i:j]) * 1024 * 2 / REPS
        wr = sum(x[2] for x in w[i:j]) * 1024 / REPS
        names = sorted(set(x[1].split("(")[0][-60:] for x in f[i:j]))
        s2 = dict(s)
        s2.update(read_bytes=round(fr), write_bytes=round(wr), read_ratio=round(fr / s["algorithmic_read_bytes"], 2),
                  write_ratio=round(wr / s["algorithmic_write_bytes"], 2), traffic_ratio=round((fr + wr) / s["algorithmic_bytes"], 2), kernels=names)
        res["shapes"].append(s2)
        i = j
    json.dump(res, open(out, "w"), indent=1)
    for s in res["shapes"]:
        print(f"{s['name']:42s} {s['M']:6d}x{s['N']:5d}x{s['K']:5d}  read {s['read_bytes'] / 1e6:8.1f} MB ({s['read_ratio']:.2f}x)  "
              f"written {s['write_bytes'] / 1e6:8.1f} MB ({s['write_ratio']:.2f}x)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        rows = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 32768
        plans = sys.argv[sys.argv.index("--plans") + 1] if "--plans" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_gemm_plans_small.json")
        run(rows, plans)
    else:
        parse(*sys.argv[2:5])
