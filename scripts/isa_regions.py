"""Static instruction mix of one kernel in a hipcc -S listing, split at s_barrier: a map of where the VALU/SALU
instructions of chain_fused_kernel sit (loops are counted once; use with the PMC totals in profiles/).

    python scripts/isa_regions.py /tmp/isa/cfk.s _ZN3gsm18chain_fused_kernelIdLi7ELb1EEEvNS_9FusedArgsE
"""
import re, sys
from collections import Counter
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith(('s_load', 's_buffer_load')): return 'smem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_')): return 'vmem'
    return 'other'
reg = Counter(); regions = []; label = None; first = start
for i in range(start + 1, end + 1):
    t = lines[i].strip()
    if not t or t.startswith((';', '.')) : continue
    if t.endswith(':'): continue
    op = t.split()[0]
    if op == 's_barrier':
        regions.append((first, i, reg)); reg = Counter(); first = i; continue
    reg[cls(op)] += 1
regions.append((first, end, reg))
keys = ['valu', 'lane', 'mfma', 'salu', 'smem', 'wait', 'br', 'lds', 'vmem']
print('%8s %8s ' % ('from', 'to') + ' '.join('%6s' % k for k in keys))
tot = Counter()
for a, b, r in regions:
    print('%8d %8d ' % (a + 1, b + 1) + ' '.join('%6d' % r[k] for k in keys)); tot.update(r)
print('%17s ' % 'total' + ' '.join('%6d' % tot[k] for k in keys))
