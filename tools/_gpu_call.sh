cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out/r3i
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3i/bench.json').read().strip().split('\n')[-1])
print(round(d['value']))
for w,v in d['workloads'].items():
    print(w, round(v['value']), v['roofline'].get('avg_launch_ms'), v['roofline'].get('launch_ms_samples'))
PY
