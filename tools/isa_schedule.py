#!/usr/bin/env python3
"""Dev tool: compress the instruction stream of one kernel in a hipcc -S listing into class runs
(MFMA / V(ALU) / dsR / dsW / gL / gS / waits / branches) to see how the compiler interleaved them.
usage: isa_schedule.py file.s mangled-name-substring [max_chars]"""
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if key in l and l.startswith('_Z') and ':' in l]
name = names[0]
a = s.index('\n' + name + ':')
b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
def cls(l):
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'): return None
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith(('ds_read', 'ds_load')): return 'dsR'
    if op.startswith(('ds_write', 'ds_store')): return 'dsW'
    if op.startswith(('global_load', 'buffer_load')): return 'gL'
    if op.startswith(('global_store', 'buffer_store')): return 'gS'
    if op.startswith('s_waitcnt'): return 'W[' + l.split(None, 1)[1].strip() + ']'
    if op.startswith('s_barrier'): return 'BAR'
    if op.startswith(('s_cbranch', 's_branch')): return 'BR'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('scratch_'): return 'SCR'
    if op.endswith('_dpp') or 'dpp' in l: return 'DPP'
    if op.startswith('v_'): return 'V'
    if op.startswith('s_'): return 's'
    return op
seq = []
for l in body:
    c = cls(l)
    if c is None:
        if l.strip().endswith(':') and l.strip().startswith('.LBB'): seq.append(('\nLABEL ' + l.strip(), 1))
        continue
    if seq and seq[-1][0] == c: seq[-1] = (c, seq[-1][1] + 1)
    else: seq.append((c, 1))
out = ' '.join(f'{c}x{n}' if n > 1 else c for c, n in seq)
print(name, len(body), 'lines')
print(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 8000])
