"""Breaks the LAST optimisation step of a rocprofv3 (rocpd sqlite) kernel trace down by kernel: python scripts/rocpd_step.py <db>"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
bursts = []
for i in idx:
    if not bursts or i - bursts[-1][-1] > 50:
        bursts.append([i])
    else:
        bursts[-1].append(i)
s, e = bursts[-2][-1] + 1, bursts[-1][-1] + 1
step = rows[s:e]
span = (step[-1][2] - step[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in step) / 1e6
print(f'last step: span {span:.1f} ms, kernel busy {busy:.1f} ms, {len(step)} kernels')
agg = collections.defaultdict(lambda: [0.0, 0])
clean = lambda n: re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0][:100]
for r in step:
    a = agg[clean(r[0])]
    a[0] += (r[2] - r[1]) / 1e6
    a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'{t:9.2f} ms {c:5d}  {n}')
gaps = sorted(((step[i + 1][1] - step[i][2]) / 1e6, i) for i in range(len(step) - 1))[-12:]
print('largest gaps (ms, after kernel):', [(round(g, 2), clean(step[i][0])[:40]) for g, i in gaps])
print('sum of positive gaps ms', sum(max(0, (step[i + 1][1] - step[i][2])) for i in range(len(step) - 1)) / 1e6)
