cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3fin3; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json'))
print('randla', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'), d['latency']['batch_1']['ms_per_frame_median'], d['latency']['batch_4']['ms_per_frame_median'])
for k,w in d['workloads'].items(): print(k, round(w.get('value',0),1), w.get('roofline',{}).get('frac'), w.get('config',{}).get('lanes'), w.get('error'))
"
