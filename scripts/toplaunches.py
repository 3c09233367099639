"""Largest individual kernel launches of the last training step in a `rocprofv3 --kernel-trace` csv: python toplaunches.py DIR [n] [name filter]."""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', '')) for r in csv.DictReader(open(f))), key=lambda r: r[0])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[2]]
ends = [ad[i] for i in range(len(ad)) if i + 1 == len(ad) or ad[i + 1] - ad[i] > 50]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = step[0][0]
print(f'{len(step)} launches, busy {sum(e - s for s, e, *_ in step) / 1e6:.1f} ms, wall {(step[-1][1] - t0) / 1e6:.1f} ms')
skip = ('igemm', 'wino23_fused')
only = sys.argv[3] if len(sys.argv) > 3 else None
for s, e, name, gx, wx in sorted(step, key=lambda r: r[0] - r[1])[:(len(step) if len(sys.argv) > 3 else n * 3)]:
    if any(k in name for k in skip) or (only and only not in name):
        continue
    print(f'{(e - s) / 1e6:7.3f} ms at +{(s - t0) / 1e6:6.1f}  grid {gx:>10} x {wx:>4}  {name[:110]}')
    n -= 1
    if n == 0:
        break
