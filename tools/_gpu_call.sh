cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_randlanet.py tests/test_gpu_configs.py -x -q -k "randla" 2>&1 | tail -3
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-overlap --no-latency --no-workloads --breakdown > gpurun_out/r4m/breakdown.json 2> gpurun_out/r4m/breakdown.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4m/breakdown.json').readline())
print(d['value'], d['ms_per_step'])
b=d['breakdown_ms']
print({k: round(b[k],3) for k in sorted(b, key=lambda k:int(k.split(':')[1])) if k.startswith('fwd:') and int(k.split(':')[1]) < 8 or k=='fwd:1000'})
print('fwd sum', sum(v for k,v in b.items() if k.startswith('fwd')))
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.0f frames/s step %.2f ms' % (d['value'], d['ms_per_step']))"
python tools/latency_only.py 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('B1', d['batch_1']['ms_per_frame_median'], 'B4', d['batch_4']['ms_per_frame_median'])"
