#!/bin/bash
# Experiment: do two detect loops on ONE GPU (two processes, each replaying its captured step) fill each other's kernel tails?
# Prints the single-process figure, then the two concurrent figures (their sum is the GPU's aggregate throughput).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/two_procs
mkdir -p $O
ARGS="--steps 60 --warmup 3 --no-train --bulk-files 0 --no-cpu-baseline --no-dense-reference"
python3 $R/bench.py $ARGS > $O/single.json 2> $O/single.err
python3 $R/bench.py $ARGS > $O/a.json 2> $O/a.err &
PA=$!
python3 $R/bench.py $ARGS > $O/b.json 2> $O/b.err &
PB=$!
wait $PA $PB
python3 - <<PY
import json
for n in ('single', 'a', 'b'):
    try:
        d = json.loads(open('$O/%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['value'], 1), 'clips/s', round(d['ms_per_step'], 2), 'ms/step; eager', round(d['eager_with_events']['ms_per_step'], 2))
    except Exception as e:
        print(n, 'failed', e)
PY
