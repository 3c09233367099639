"""Summarise rocprofv3 --pmc csv output: per kernel name and counter, mean value per dispatch.  usage: pmc_summarize.py DIR [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(f'{k:70s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}')
