cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3fin5; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json'))
print('randla', round(d['value'],1), d['step_ms_median'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('avg_launch_ms_alone'), d['latency']['batch_1']['ms_per_frame_median'], d['latency']['batch_4']['ms_per_frame_median'])
for k,w in d['workloads'].items(): print(k, round(w.get('value',0),1), w.get('step_ms_median'), w.get('roofline',{}).get('frac'), w.get('error'))
"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/pf -o rl -- python $R/bench.py --no-workloads --no-latency --no-cpu-baseline > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $O/pf/rl_results.db > $O/rl_kernel_stats.csv 2>/dev/null
rm -rf $O/pf
