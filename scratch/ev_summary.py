"""Prints the numbers of gpurun_out/ev (scratch/refresh_evidence.sh) that DESIGN.md section 6 quotes."""
import json, sys
d0 = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ev"
def L(p):
    return [json.loads(l) for l in open(p) if l.startswith('{')]
d = L(f'{d0}/r06_bench_e2e.json')[0]
r = d['roofline']
print('e2e', d['value'], d['ms_per_step'], 'frame', r['ms_per_step'], r['frac'], r.get('frac_moved_of_achievable'), r.get('achieved'), 'conv', d['cnn']['conv_time_ms_per_step'], d['cnn']['samples_per_s'], d['cnn']['conv_tflops_all_layers'])
print(' layers', d['cnn']['layers_ms_per_step'])
for k, v in r['kernels'].items():
    print(' ', k, v['ms_per_step'], v['mfma']['achieved'], v['mfma']['frac'], 'hbm', v['hbm']['achieved'], 'traffic/alg', round(v['traffic'] / v['algorithmic_bytes_per_launch'], 3) if v.get('traffic') else None, 'us', v['avg_launch_us'])
print(' agg', {k: v for k, v in r['conv_aggregate'].items() if k != 'note'})
for k in ('default_config', 'fs64', 'cpu_baseline', 'bf16x2', 'bf16x3'):
    v = d.get(k)
    if isinstance(v, dict): print(k, {kk: vv for kk, vv in v.items() if not isinstance(vv, (dict, list)) and kk not in ('what', 'note', 'sample')})
ff = d.get('from_files', {})
print('from_files', {k: v for k, v in ff.items() if not isinstance(v, (dict, list)) and k != 'what'})
print(' split', ff.get('split_s'))
fx = ff.get('fixture_recordings') or {}
print('fixture', {k: v for k, v in fx.items() if not isinstance(v, (dict, list)) and k != 'what'})
print(' split', fx.get('split_s'))
for f in ('f32math', 'bf16x3', 'track', 'config4', 'ir'):
    x = L(f'{d0}/r06_bench_{"e2e_" if f in ("f32math", "bf16x3") else ""}{f}.json')[0]
    print(f, x['value'], x['ms_per_step'], x.get('roofline', {}).get('frac'), x.get('roofline', {}).get('achieved'))
for l in L(f'{d0}/r06_directory_bulk.json'):
    print('dir', {k: v for k, v in l.items() if not isinstance(v, (dict, list))})
