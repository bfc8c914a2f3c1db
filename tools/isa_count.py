#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from hipcc -S output: tools/isa_count.py file.s kernel_substring [depth]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
name = [l for l in re.findall(r'\.globl\t(\S+)', s) if key in l][0]
i = s.index(name + ':')
j = s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)
j = s.index('s_endpgm', i)
lines = s[i:j].split('\n')
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_'): return 'salu'
    return 'other'
# segment by labels
seg, cur, name_ = [], collections.Counter(), 'entry'
ops = collections.Counter()
for l in lines:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'): continue
    m = re.match(r'^(\.LBB\S+):\s*(;.*)?$', t)
    if m:
        seg.append((name_, cur)); cur = collections.Counter(); name_ = m.group(1) + ' ' + (m.group(2) or '')
        continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if cls(op) == 'valu': ops[op] += 1
seg.append((name_, cur))
tot = collections.Counter()
for n, c in seg:
    if sum(c.values()) >= int(sys.argv[3]) if len(sys.argv) > 3 else 40:
        print(f"{n[:70]:70s}", dict(c))
    tot.update(c)
print('TOTAL', dict(tot))
print(ops.most_common(25))
