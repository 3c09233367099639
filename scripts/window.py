"""Kernels of the last training step of a rocprofv3 --kernel-trace csv inside [t0, t1] ms after the step's start: python window.py DIR t0 t1"""
import csv, glob, sys
d, w0, w1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[2]]
ends = [ad[i] for i in range(len(ad)) if i + 1 == len(ad) or ad[i + 1] - ad[i] > 50]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = step[0][0]
prev = None
n = busy = 0
for s, e, name in step:
    ts = (s - t0) / 1e6
    if w0 <= ts <= w1:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f'+{ts:8.3f} ms  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {name[:100]}')
        n += 1; busy += e - s
    prev = max(prev, e) if prev is not None else e
print(f'{n} kernels, busy {busy / 1e6:.2f} ms of {w1 - w0:.1f} ms')
