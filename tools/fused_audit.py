"""Audit of csrc/attention_bwd_fused.hip's ISA (hipcc -S output): inside the slice loop, (1) no scratch access (a spill reload is a
vector-memory operation: it breaks the counted s_waitcnt vmcnt the loop relies on), and (2) no vector-memory load with a register destination is left in flight
past its own statement (hipcc counts an asm load's destination as written at the end of the statement and may copy it at any point
after that: round 4's stale lse / delta under load was such a copy, placed right in front of the wait that covered the load).    python tools/fused_audit.py /tmp/t/f/fused.s"""
import re
import sys

s = open(sys.argv[1]).read()
bad = 0


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


for name in re.findall(r'^(_ZN\S*attn_bwd_fused_kernel\S*?):', s, re.M):
    i = s.index(name + ':')
    lines = s[i:s.index('.Lfunc_end', i)].split('\n')
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    for n, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if not (m and labels.get(m.group(1), 1 << 30) < n):
            continue
        body = lines[labels[m.group(1)]:n]
        if sum('v_mfma' in x for x in body) != 80:
            continue
        nscr = sum('scratch_' in x for x in body)
        print(f"{name[:70]}: loop of {len(body)} lines, scratch accesses {nscr}")
        bad += nscr
        # every vector-memory load with a REGISTER destination is waited for by the very next instruction (its own asm statement):
        # the register allocator may copy such a destination anywhere after the statement, so nothing may stay in flight past it
        for k, x in enumerate(body):
            if re.search(r'\blds\b|_load_lds_', x) or not re.search(r'(global_load_\w+ v|buffer_load_\w+ v|flat_load_\w+ v)', x):
                continue
            nxt = next(y.strip() for y in body[k + 1:] if y.strip() and not y.strip().startswith(';'))
            if not re.match(r's_waitcnt vmcnt\(0\)', nxt):
                print(f"   register-destination load left in flight: {x.strip()}  (next: {nxt})")
                bad += 1
        break
print("audit", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
