#!/usr/bin/env python3
"""Where a kernel's register spills sit: scripts/spill_map.py <file.s> <mangled kernel name prefix>
Lists the basic blocks of the kernel that hold scratch stores / loads with their loop depth (from the compiler's asm comments)."""
import re, sys
src, name = sys.argv[1], sys.argv[2]
text = open(src).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith(name) and l.rstrip().endswith(':') or (l.startswith(name) and ':' in l))
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
blocks, cur = [], {'name': 'entry', 'line': start, 'st': 0, 'ld': 0, 'n': 0, 'depth': 0}
blocks.append(cur)
for i in range(start + 1, end):
    l = text[i]
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur = {'name': m.group(1), 'line': i, 'st': 0, 'ld': 0, 'n': 0, 'depth': 0}
        d = re.search(r'Depth=(\d+)', m.group(2))
        if d: cur['depth'] = int(d.group(1))
        blocks.append(cur); continue
    if l.strip().startswith(';'):
        d = re.search(r'Depth[= ](\d+)', l)
        if d and cur['n'] == 0: cur['depth'] = max(cur['depth'], int(d.group(1)))
        continue
    if 'scratch_store' in l: cur['st'] += 1
    if 'scratch_load' in l: cur['ld'] += 1
    if l.strip(): cur['n'] += 1
by = {}
for b in blocks:
    if b['st'] or b['ld']:
        print(f"{b['name']:12s} asm line {b['line'] - start:5d} depth {b['depth']} insts {b['n']:4d} stores {b['st']:3d} loads {b['ld']:3d}")
        k = by.setdefault(b['depth'], [0, 0]); k[0] += b['st']; k[1] += b['ld']
print("by loop depth (stores, loads):", dict(sorted(by.items())))
