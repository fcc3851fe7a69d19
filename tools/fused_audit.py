"""Audit of csrc/attention_bwd_fused.hip's ISA (hipcc -S output): inside the slice loop, (1) no scratch access (a spill reload is a
vector-memory operation: it breaks the counted s_waitcnt vmcnt the loop relies on), and (2) no instruction touches the destination
registers of an asm-issued load between the load and the wait that covers it (hipcc counts an asm load's destination as written at
the end of the statement).    python tools/fused_audit.py /tmp/t/f/fused.s"""
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
        for k, x in enumerate(body):
            if 'lds' in x or not re.search(r'(global_load_dword\w* v|buffer_load_dword\w* v)', x):
                continue
            dst = regs(x.split(',')[0])
            for kk in range(k + 1, k + len(body)):
                y = body[kk % len(body)].strip()
                if re.match(r's_waitcnt vmcnt\(\d+\)', y):
                    break
                if y and not y.startswith(';') and regs(y) & dst:
                    print(f"   TOUCHED before its wait: {x.strip()}  <-  {y}")
                    bad += 1
        break
print("audit", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
