"""Instruction classes of one kernel's loops in a hipcc -S listing, one line per loop: where the spills, waits and MFMAs are.
    python tools/isa_slots.py file.s <kernel name substring> [--dump LABEL]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
for k in re.split(r'\n(?=_Z\w+:[^\n]*\n)', s):
    m = re.match(r'(_Z\w+):', k)
    if not m or want not in m.group(1):
        continue
    lines = k.split('\n')
    labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'(\.LBB\d+_\d+):', l)] if mm}
    print(m.group(1))
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and labels.get(mm.group(1), 1 << 30) < i:
            a = labels[mm.group(1)]
            body = lines[a:i]
            cnt = lambda pat: sum(1 for x in body if re.search(pat, x))
            if cnt('v_mfma') == 0:
                continue
            print(f"  loop {mm.group(1)} [{a}-{i}] {i - a} lines: mfma {cnt('v_mfma')} accvgpr {cnt('v_accvgpr')} scratch {cnt('scratch_')} ds_read {cnt('ds_read')} "
                  f"ds_write {cnt('ds_write')} exp {cnt('v_exp')} waitcnt {cnt('s_waitcnt')} (vmcnt {cnt('vmcnt')}) nop {cnt('s_nop')} dma {cnt('buffer_load')} "
                  f"global {cnt('global_')} branch {cnt('s_cbranch')}")
            if dump == mm.group(1):
                def cls(x):
                    t = x.split()
                    if not t or t[0].startswith(';') or t[0].startswith('.'):
                        return None
                    o = t[0]
                    if o.startswith('v_mfma'): return 'MFMA'
                    if o.startswith('ds_read'): return 'r'
                    if o.startswith('ds_write'): return 'W'
                    if o == 's_waitcnt': return 'wait(' + ' '.join(t[1:]) + ')'
                    if o == 's_nop': return 'nop'
                    if o == 's_barrier': return 'BARRIER'
                    if o.startswith('buffer_load'): return 'DMA'
                    if o.startswith('global_') or o.startswith('scratch_'): return o
                    if o.startswith('v_exp'): return 'e'
                    if o.startswith('v_accvgpr'): return 'acc'
                    if o.startswith('v_'): return 'v'
                    if o.startswith('s_cbranch'): return 'BR'
                    if o.startswith('s_'): return 's'
                    return o
                seq, out = [c for c in map(cls, body) if c], []
                for c in seq:
                    if out and out[-1][0] == c: out[-1][1] += 1
                    else: out.append([c, 1])
                txt = ' '.join((f"{n}{c}" if n > 1 else c) for c, n in out)
                print(re.sub(r' (MFMA)', r'\n   \1', txt))
    break
