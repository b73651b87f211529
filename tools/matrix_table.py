#!/usr/bin/env python3
"""matrix_table.py <matrix.log of a tools/r06*.sh session> -- the session's runs side by side (product and A/B builds alternate in one session: boxes differ by +-1.5 %)"""
import re, sys
rows = {}; cur = None; keys = []
for ln in open(sys.argv[1]):
    if ln.startswith('=='):
        cur = re.sub(r' \d$', '', ln.strip('= \n'))
        if cur not in keys: keys.append(cur)
        continue
    m = re.match(r'(\w+): ', ln)
    if m: w = m.group(1); continue
    m = re.search(r'([\d.]+) ms\s+([\d.]+) GB/s', ln)
    if m: rows.setdefault(w, {}).setdefault(cur, []).append((float(m.group(1)), float(m.group(2))))
print('%-10s ' % '' + ' '.join('%26s' % k[:26] for k in keys))
for w, r in rows.items():
    one = w in ('appf1', 'book1')
    print('%-10s ' % w + ' '.join('%26s' % (' / '.join(('%.3f' % a) if one else ('%.1f' % b) for a, b in r.get(k, [])) + (' ms' if one else ' GB/s')) for k in keys))
