"""Compress a kernel's ISA (hipcc -save-temps .s file) into a trace of memory ops, waits, MFMAs and VALU run
lengths — shows how many dependent memory round trips a loop body has.
usage: isa_trace.py file.s mangled_kernel_name [start_label [end_label]]"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
if len(sys.argv) > 3:
    a = [k for k, l in enumerate(body) if l.startswith(sys.argv[3] + ':')][0]
    b = [k for k, l in enumerate(body) if l.startswith(sys.argv[4] + ':')][0] if len(sys.argv) > 4 else len(body)
    body = body[a:b]
out, run, kind, nv = [], 0, None, 0
def flush():
    global run, kind
    if run: out.append(kind if run == 1 else f"{kind}x{run}")
    run, kind = 0, None
for ln in body:
    t = ln.strip(); k = None
    if t.startswith(('global_load', 'buffer_load')): k = 'GL' + ('128' if 'dwordx4' in t else '96' if 'dwordx3' in t else '64' if 'dwordx2' in t else '32')
    elif t.startswith('global_store'): k = 'GS'
    elif t.startswith('s_load'): k = 'SL'
    elif t.startswith(('ds_read', 'ds_load')): k = 'dr'
    elif t.startswith(('ds_write', 'ds_store')): k = 'dw'
    elif t.startswith('v_mfma'): k = 'MFMA'
    elif t.startswith('s_waitcnt'):
        m = re.search(r'vmcnt\((\d+)\)', t); m2 = re.search(r'lgkmcnt\((\d+)\)', t)
        k = ('Wvm(%s)' % m.group(1)) if m else (('Wlgkm(%s)' % m2.group(1)) if m2 else None)
    elif t.startswith('s_barrier'): k = 'BARRIER'
    elif re.match(r'\.LBB\d+_\d+:', t): k = '\n[' + t.split(':')[0] + ']'
    elif t.startswith(('s_cbranch', 's_branch')): k = '<' + t.split()[0][2:] + ' ' + t.split()[-1] + '>'
    elif t.startswith('v_'): nv += 1; continue
    else: continue
    if k is None: continue
    if nv: flush(); out.append(f"v{nv}"); nv = 0
    if k == kind: run += 1
    else: flush(); kind = k; run = 1
flush()
print(' '.join(out))
