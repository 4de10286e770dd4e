"""Per-basic-block instruction-class counts of one kernel in a hipcc -S file (valu / mfma / vector loads / stores / LDS / scalar), with backward branches marked as loops;
with a third argument also the opcode histogram of its vector instructions.  usage: isa_blocks.py file.s mangled_kernel_name [ops]
(how the per-view and per-tile instruction counts of DESIGN.md 5 were read off the ISA; tools/isa_trace.py gives the load / wait / MFMA skeleton)"""
import re, sys, collections
s = open(sys.argv[1]).read(); name = sys.argv[2]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
lines = s[i:j].split('\n')
blocks = []; cur = ['entry', collections.Counter(), []]
order = {}
for ln in lines:
    t = ln.strip()
    m = re.match(r'(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), collections.Counter(), []]; continue
    if not t or t.startswith((';', '.', '_Z')): continue
    op = t.split()[0]
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith(('global_load', 'buffer_load')): k = 'vld'
    elif op.startswith(('global_store', 'buffer_store', 'global_atomic')): k = 'vst'
    elif op.startswith('ds_'): k = 'lds'
    elif op.startswith('s_load') or op.startswith('s_buffer_load'): k = 'smem'
    elif op.startswith('s_waitcnt'): k = 'wait'
    elif op.startswith(('s_cbranch', 's_branch')): k = 'br'; cur[2].append(t.split()[-1])
    elif op.startswith('s_'): k = 'salu'
    else: k = 'other'
    cur[1][k] += 1
    if len(sys.argv) > 3 and k == 'valu': cur[1]['op:' + op] += 1
blocks.append(cur)
idx = {b[0]: n for n, b in enumerate(blocks)}
tot = collections.Counter()
for n, (lab, c, br) in enumerate(blocks):
    back = [t for t in br if t in idx and idx[t] <= n]
    tot.update({k: v for k, v in c.items() if not k.startswith('op:')})
    print(f"{n:3d} {lab:12s} valu {c['valu']:5d} mfma {c['mfma']:3d} vld {c['vld']:3d} vst {c['vst']:3d} lds {c['lds']:3d} smem {c['smem']:3d} salu {c['salu']:4d} wait {c['wait']:3d}  -> {' '.join(br)} {'LOOP->' + ','.join(back) if back else ''}")
print('total', dict(tot))
if len(sys.argv) > 3:
    ops = collections.Counter()
    for b in blocks:
        for k, v in b[1].items():
            if k.startswith('op:'): ops[k[3:]] += v
    for k, v in ops.most_common(60): print(f"  {k:28s} {v}")
