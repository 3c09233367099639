import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    t = float(r['TotalDurationNs']) / 1e6 / n; tot += t
    if t > 0.08: print('%-120s %6.1f calls %7.3f ms' % (r['Name'][:120], int(r['Calls']) / n, t))
print('total', tot)
