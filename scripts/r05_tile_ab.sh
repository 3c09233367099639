mkdir -p gpurun_out/r05
for t in 8 0 8 0; do
  python bench.py --tile-clips $t --no-train --bulk-files 0 --no-cpu-baseline --no-dense-reference --no-split-leg > gpurun_out/r05/b_$t.json 2>/dev/null
  python - $t <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/r05/b_{t}.json").read().strip().splitlines()[-1])
print(f"tile {t}:", round(d["value"], 1), round(d["ms_per_step"], 2), "single", round(d["single_lane_graph_replay"]["ms_per_step"], 2), "eager", round(d["eager_with_events"]["ms_per_step"], 2),
      "dets/step", d["config"]["detections_per_step"], "dominant ms/step", round(d["roofline"]["ms_per_step"], 2), "gemm-type", round(d["roofline"]["all_gemm_type_launches_ms_per_step"], 2))
PY
done
