cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3fin7; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json'))
print('randla', round(d['value'],1), d['step_ms_median'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('avg_launch_ms_alone'))
for k,w in d['workloads'].items(): print(k, round(w.get('value',0),1), w.get('step_ms_median'), w.get('roofline',{}).get('frac'), w.get('error'))
"
