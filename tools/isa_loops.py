"""Where a kernel's scratch traffic, MFMAs and waits sit: tools/isa_loops.py file.s mangled-substring
Prints, per basic block label, the counts of v_mfma / ds_read / scratch_ / s_waitcnt / s_barrier instructions."""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\S+):', s, re.M) if key in n]
for name in names:
    body = s[s.index(name + ':'):]
    body = body[:body.index('s_endpgm')]
    print('==', name[:110])
    cur, stats, order = 'entry', {}, []
    for l in body.split('\n'):
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            cur = m.group(1)
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        d = stats.setdefault(cur, dict(n=0, mfma=0, ds=0, scr=0, wait=0, bar=0, br=''))
        if cur not in order: order.append(cur)
        d['n'] += 1
        if t.startswith('v_mfma'): d['mfma'] += 1
        if t.startswith('ds_'): d['ds'] += 1
        if t.startswith('scratch_'): d['scr'] += 1
        if t.startswith('s_waitcnt'): d['wait'] += 1
        if t.startswith('s_barrier'): d['bar'] += 1
        if t.startswith('s_cbranch') or t.startswith('s_branch'): d['br'] += t.split()[-1] + ' '
    for k in order:
        d = stats[k]
        if d['mfma'] or d['scr'] or d['bar']:
            print(f"  {k:14s} instr {d['n']:5d} mfma {d['mfma']:3d} ds {d['ds']:3d} scratch {d['scr']:3d} waitcnt {d['wait']:3d} barrier {d['bar']} -> {d['br']}")
