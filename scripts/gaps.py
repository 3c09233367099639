"""Idle gaps of the GPU in the last training step of a rocprofv3 --kernel-trace csv: python gaps.py DIR [min_gap_us]."""
import csv, glob, sys
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# last step = after the last adamw_kernel burst but one
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[2]]
ends = [ad[i] for i in range(len(ad)) if i + 1 == len(ad) or ad[i + 1] - ad[i] > 50]
lo, hi = ends[-2] + 1, ends[-1]
step = rows[lo:hi + 1]
t0, t1 = step[0][0], step[-1][1]
busy = sum(e - s for s, e, _ in step)
print(f'last step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms')
gaps = []
cur_end = step[0][1]
for i in range(1, len(step)):
    g = step[i][0] - cur_end
    if g > thr * 1e3:
        gaps.append((g, i))
    cur_end = max(cur_end, step[i][1])
tot = 0
for g, i in sorted(gaps, reverse=True)[:25]:
    tot += g
    print(f'{g / 1e6:7.2f} ms idle at +{(step[i][0] - t0) / 1e6:7.1f} ms  after {step[i - 1][2][:60]:60s} before {step[i][2][:60]}')
print(f'gaps > {thr} us: {len(gaps)}, total {sum(g for g, _ in gaps) / 1e6:.1f} ms')
small = sum(max(0, step[i][0] - max(s[1] for s in step[max(0, i - 3):i])) for i in range(1, len(step)))
