cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4n
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err
tail -c 600 gpurun_out/r4n/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4n/bench.json').readline())
print(d['value'], d['unit'], d['ms_per_step'], d.get('ranks_seen'))
print('roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if not isinstance(v,(dict,list,str))})
print('cpu', d['cpu_baseline'])
for w in ('kpconv','pointpillars'):
    x=d.get('workloads',{}).get(w)
    if x: print(w, x['value'], x['unit'], x['ms_per_step'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in x['roofline'].items() if k in ('frac','frac_alone','avg_launch_ms','avg_launch_ms_alone','achieved','end_to_end_tflops','traffic')})
print('latency', d.get('latency'))
PY
