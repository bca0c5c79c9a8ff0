# GPU box, one call: parity suite on the product library, the bench line, then product vs probe builds (A/B).
#   gpurun --timeout 1500 -- 'bash scripts/run_round_checks.sh <tag> "<kinds>" <probe names...>'
TAG=${1:-r04a}; KINDS=${2:-"stft mel"}; shift 2
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -5 gpurun_out/pytest_$TAG.log
timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$TAG.json'))
    print('value %.1f M  ms %.4f  rep %s' % (d['value'] / 1e6, d['ms_per_step'], [round(x, 4) for x in d['repeats']['ms_per_step_all']]))
    for k in ('roofline_stft', 'roofline_istft'):
        print(k, round(d[k]['launch_ms'], 4), round(d[k]['frac'], 4), d[k].get('round_trip_snr_db_min'))
    print('stream', json.dumps(d.get('stream_ceiling'))[:600])
    print('power', json.dumps(d.get('board_power'))[:600])
    print('cpu', json.dumps(d.get('cpu_baseline'))[:500]); print('cpu_all', json.dumps(d.get('cpu_baseline_all_cores'))[:300]); print('parity', d.get('parity'))
    print('cqt', json.dumps(d.get('cqt_lite'))[:900]); print('e2e', json.dumps(d.get('end_to_end_numpy'))[:400]); print('gl', json.dumps(d.get('griffinlim'))[:200])
except Exception as e:
    print('bench parse failed', e)
PY
if [ $# -gt 0 ]; then timeout 600 bash scripts/ab_run.sh "$KINDS" "$@" 2>&1 | tail -40; fi
