"""Per-kernel breakdown of the LAST window of a rocprofv3 (rocpd sqlite) trace that ends with a marker kernel:
python scripts/rocpd_window.py <db> <marker substring> [top]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2]
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
s, e = idx[-2] + 1, idx[-1] + 1
step = rows[s:e]
span = (step[-1][2] - step[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in step) / 1e6
print(f'window: span {span:.2f} ms, kernel busy {busy:.2f} ms, {len(step)} kernels')
clean = lambda n: re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0][:100]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in step:
    a = agg[clean(r[0])]
    a[0] += (r[2] - r[1]) / 1e6
    a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f'{t:9.3f} ms {c:5d}  {n}')
