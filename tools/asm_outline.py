"""Outline of one kernel in a hipcc -S listing: labels, branches, barriers, waits, LDS-DMA, scratch (spill) traffic, MFMA runs.
    python tools/asm_outline.py /tmp/tile.s ILi5ELi0ELi2E [max_lines]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 400
m = re.search(r'\n(_ZN\S*' + re.escape(pat) + r'\S*):.*?\n(.*?)\n\.Lfunc_end', txt, re.S)
name, body = m.group(1), m.group(2)
out, cnt = [], 0
for i, l in enumerate(body.split('\n')):
    if 'v_mfma' in l:
        cnt += 1
        continue
    if cnt:
        out.append(f'        [{cnt} mfma]')
        cnt = 0
    if re.search(r'scratch_|s_barrier|^\.LBB|s_cbranch|s_waitcnt vmcnt|load_lds|lds$|global_load|global_store|ds_read|ds_write', l):
        out.append(f'{i:6d}: {l.strip()[:110]}')
# compress runs of same-kind lines
res, last, run = [], None, 0
for l in out:
    k = re.sub(r'[0-9]+', '', l.split(':', 1)[-1].strip().split(' ')[0])
    if k == last and k in ('ds_read_b', 'ds_write_b', 'global_load_dwordx', 'global_store_dwordx', 'global_load_lds_dwordx'):
        run += 1
        continue
    if run:
        res.append(f'        ... +{run} more {last}')
    res.append(l)
    last, run = k, 0
print(name)
print('\n'.join(res[:mx]))
