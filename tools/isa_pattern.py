"""Instruction pattern of a kernel's basic blocks that contain MFMAs: tools/isa_pattern.py file.s mangled-substring [min_mfma]
M = v_mfma, r = ds_read, W = ds_write, DMA = buffer_load..lds, GL / GS = global load / store, w[..] = s_waitcnt, BAR = s_barrier,
(n) = n other instructions."""
import re, sys
s = open(sys.argv[1]).read(); key = sys.argv[2]; minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
name = [n for n in re.findall(r'^(_Z\S+):', s, re.M) if key in n][0]
body = s[s.index(name + ':'):]; body = body[:body.index('s_endpgm')]
blocks, cur = {}, 'entry'
for l in body.split('\n'):
    m = re.match(r'^(\.LBB\S+):', l)
    if m: cur = m.group(1)
    blocks.setdefault(cur, []).append(l)
print('==', name[:120])
for k, ls in blocks.items():
    if sum('v_mfma' in x for x in ls) < minm: continue
    out = []
    for l in ls:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'): continue
        op = t.split()[0]
        if op.startswith('v_mfma'): out.append('M')
        elif op.startswith('ds_read'): out.append('r')
        elif op.startswith('ds_write'): out.append('W')
        elif op.startswith('buffer_load') and ' lds' in t: out.append('DMA')
        elif op.startswith('global_load') or op.startswith('buffer_load'): out.append('GL')
        elif op.startswith('global_store') or op.startswith('buffer_store'): out.append('GS')
        elif op == 's_waitcnt': out.append('w[' + t.split(None, 1)[1].replace('lgkmcnt', 'L').replace('vmcnt', 'V') + ']')
        elif op.startswith('s_barrier'): out.append('BAR')
        elif op.startswith('scratch_'): out.append('SCR')
        else: out.append('.')
    res, i = [], 0
    while i < len(out):
        if out[i] == '.':
            j = i
            while j < len(out) and out[j] == '.': j += 1
            res.append(f'({j - i})'); i = j
        elif out[i] in ('M', 'r'):
            j = i
            while j < len(out) and out[j] == out[i]: j += 1
            res.append(out[i] + (str(j - i) if j - i > 1 else '')); i = j
        else: res.append(out[i]); i += 1
    print(' ', k, ':', ' '.join(res))
