#!/usr/bin/env python3
"""Static instruction mix per loop of a kernel: scripts/loop_mix.py <file.s> <mangled kernel name prefix> [min LDS instructions]
Uses the compiler's asm comments ("=>This Loop Header: Depth=N", "in Loop: Header=BBx_y Depth=N") to attribute every basic block to
its innermost loop; prints LDS loads / stores / atomics, VALU, SALU, scratch and barriers per loop (blocks of nested loops are
counted in their own loop only)."""
import re, sys, collections
src, name = sys.argv[1], sys.argv[2]
min_ds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
text = open(src).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith(name) and ':' in l)
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
loops = collections.OrderedDict()
cur_loop, cur_blk = None, None
pending_hdr = None
for i in range(start + 1, end):
    l = text[i]
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur_blk = m.group(1)[2:]
        cur_loop = None
        rest = m.group(2)
        if 'Loop Header' in rest: cur_loop = (cur_blk, int(re.search(r'Depth=(\d+)', rest).group(1)))
        mm = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', rest)
        if mm: cur_loop = (mm.group(1), int(mm.group(2)))
        continue
    s = l.strip()
    if s.startswith(';'):
        if cur_loop is None:
            if 'Loop Header' in s and 'Depth=' in s: cur_loop = (cur_blk, int(re.search(r'Depth=(\d+)', s).group(1)))
            mm = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', s)
            if mm: cur_loop = (mm.group(1), int(mm.group(2)))
        continue
    if not s or s.startswith('.') or cur_loop is None: continue
    d = loops.setdefault(cur_loop, collections.Counter(line=i - start))
    op = s.split()[0]
    if op.startswith('ds_read') or op.startswith('ds_load'): d['ds_rd'] += 1
    elif op.startswith('ds_write') or op.startswith('ds_store'): d['ds_wr'] += 1
    elif op.startswith('ds_'): d['ds_other'] += 1
    elif op.startswith('v_'): d['valu'] += 1; d['f64'] += ('f64' in op)
    elif op.startswith('s_barrier'): d['barrier'] += 1
    elif op.startswith('s_waitcnt'): d['waitcnt'] += 1
    elif op.startswith('s_'): d['salu'] += 1
    elif op.startswith('scratch_'): d['scratch'] += 1
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): d['vmem'] += 1
for (h, depth), d in loops.items():
    if d['ds_rd'] + d['ds_wr'] + d['ds_other'] < min_ds: continue
    print(f"loop {h:12s} depth {depth} @+{d['line']:6d}: lds rd {d['ds_rd']:4d} wr {d['ds_wr']:4d} other {d['ds_other']:3d} | valu {d['valu']:5d} (f64 {d['f64']:4d}) salu {d['salu']:4d} waitcnt {d['waitcnt']:3d} scratch {d['scratch']:3d} vmem {d['vmem']:3d} barrier {d['barrier']}")
