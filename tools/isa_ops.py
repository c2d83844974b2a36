#!/usr/bin/env python3
"""Condensed instruction stream of one kernel from a --save-temps .s file: waits, barriers, DMA, MFMA, LDS and memory ops
with run lengths.  usage: isa_ops.py file.s kernel-substring [max lines]"""
import re, sys
src, pat = open(sys.argv[1]).read().split('\n'), sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = next(i for i, l in enumerate(src) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start, len(src)) if '.end_amdhsa_kernel' in src[i] or src[i].startswith('.Lfunc_end'))
out, last, cnt = [], None, 0
keep = re.compile(r'^\s+(s_waitcnt|s_barrier|global_load_lds|v_mfma|ds_read|ds_write|global_load|global_store|buffer_|s_cbranch|s_branch|scratch_)')
for l in src[start:end]:
    if re.match(r'^\.LBB', l):
        tok = l.split(':')[0]
    elif keep.match(l):
        f = l.split()
        tok = ' '.join(f[:4]) if f[0] == 's_waitcnt' else (f[0] + (' ' + f[1] if f[0].startswith('s_cbranch') or f[0] == 's_branch' else ''))
    else:
        continue
    if tok == last: cnt += 1
    else:
        if last: out.append(last + (f' x{cnt}' if cnt > 1 else ''))
        last, cnt = tok, 1
out.append(last + (f' x{cnt}' if cnt > 1 else ''))
print('\n'.join(out[:lim]))
