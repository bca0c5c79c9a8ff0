python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02e.log 2>&1; tail -3 gpurun_out/pytest_r02e.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r02c.json'))
print('value', d['value']/1e6, 'ms', d['ms_per_step'], 'rep', d['repeats']['ms_per_step_all'])
for k in ('roofline_stft','roofline_istft'): print(k, d[k]['launch_ms'], d[k]['frac'])
print('e2e', d.get('end_to_end_numpy')); print('gl', d.get('griffinlim')); print('cqt', d.get('cqt_lite')); print('torch', d.get('dropin_torch'))
"
