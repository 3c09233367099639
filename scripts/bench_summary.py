"""Prints the main figures of a bench.py JSON line.  usage: python scripts/bench_summary.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('detect', round(d['value'], 1), 'clips/s', round(d['ms_per_step'], 2), 'ms; distinct clips', d['config'].get('distinct_clips'), 'lanes', d['config'].get('lanes'))
r = d['roofline']
print('roofline frac', round(r['frac'], 3), 'alg bytes/launch', r.get('algorithmic_bytes_per_launch'), 'traffic', r.get('traffic'), 'wasted', r.get('wasted_traffic_ratio'),
      'whole-step frac', round(r['whole_step_executed_frac_of_mfma_peak'], 3))
print('eager', d.get('eager_with_events', {}).get('ms_per_step'), 'single', d.get('single_lane_graph_replay'), 'multi', d.get('multi_lane_graph_replay'))
s = d.get('split_bf16')
print('split detect', None if not s else {k: s.get(k) for k in ('value', 'ms_per_step', 'launch_note')}, None if not s or not s.get('roofline') else round(s['roofline']['frac'], 3))
b = d.get('bulk_inference')
print('bulk', None if not b else {k: b.get(k) for k in ('value', 'files_per_gpu', 'wall_s', 'ratio_to_resident_hbm_headline', 'error')})
t = d.get('train_step') or {}
print('train', t.get('value'), t.get('ms_per_step'), 'frac', t.get('executed_frac_of_mfma_peak'), 'split', t.get('split_bf16'), t.get('error'))
print('train with fe', t.get('with_front_end'), 'mix', t.get('reference_schedule_9_to_1') or t.get('mix'))
for k in ('roofline_backward', 'roofline_backward_data_gradients'):
    v = t.get(k) or {}
    print(k, 'frac', v.get('frac'), 'ms/step', v.get('all_wgrad_ms_per_step') or v.get('all_dgrad_ms_per_step'), 'alg bytes', v.get('algorithmic_bytes_per_launch'), 'traffic', v.get('traffic'))
c = d.get('cpu_baseline') or {}
print('cpu', c.get('value'), c.get('cores'), c.get('threads8'))
