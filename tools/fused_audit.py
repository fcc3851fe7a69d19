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
        # vector-memory loads with a REGISTER destination, two kinds:
        #  - issued by hipcc itself (outside ASMSTART / ASMEND): it tracks them and waits before their first use — safe; its count
        #    ignores what asm statements have in flight, which can only make the wait stricter
        #  - issued by an asm statement: hipcc counts the destination as written when the statement ends.  Either the statement
        #    waits itself (vmcnt(0) right behind the load), or — the rotation entries of a finished dQ tile — the load is released
        #    by a later asm s_waitcnt vmcnt whose outputs every use hangs on: then NOTHING between the load and that wait (in
        #    program text: branches taken into account by scanning every line up to the first asm wait that lists the registers'
        #    block, conservatively the first asm s_waitcnt vmcnt(0) behind the load) may read or write the destination registers
        in_asm, k = False, 0
        while k < len(body):
            x = body[k]
            if 'ASMSTART' in x: in_asm = True
            if 'ASMEND' in x: in_asm = False
            is_load = (not re.search(r'\blds\b|_load_lds_', x)) and re.search(r'(global_load_\w+ v|buffer_load_\w+ v|flat_load_\w+ v)', x)
            if not (is_load and in_asm):
                k += 1
                continue
            # the loads of this statement
            dests, j = set(), k
            while 'ASMEND' not in body[j]:
                m2 = re.search(r'_load_\w+\s+(v\[\d+:\d+\]|v\d+)', body[j])
                if m2: dests |= regs(m2.group(1))
                waits_itself = bool(re.search(r's_waitcnt vmcnt\(0\)', body[j]))
                j += 1
            if waits_itself:
                k = j
                continue
            # scan to the releasing wait
            released, touched = False, []
            in2 = False
            for y in body[j:]:
                if 'ASMSTART' in y: in2 = True
                if 'ASMEND' in y: in2 = False
                t = y.strip()
                if not t or t.startswith(';') or t.startswith('.'):
                    continue
                if in2 and re.match(r's_waitcnt vmcnt\(0\)', t):
                    released = True
                    break
                ops = t.split(None, 1)
                if len(ops) > 1 and regs(ops[1]) & dests:
                    touched.append(t)
            if not released:
                print(f"   asm load never released by an asm s_waitcnt vmcnt(0): {x.strip()}")
                bad += 1
            for t in touched:
                print(f"   destination of an asm load in flight is touched before its wait: {t}")
                bad += 1
            print(f"   asm-issued loads into v{sorted(dests)[0]}..v{sorted(dests)[-1]} ({len(dests)} registers): released by an asm wait, {len(touched)} touches in between")
            k = j
        break
print("audit", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
