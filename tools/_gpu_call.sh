cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/pytest.log; tail -4 gpurun_out/r4a/pytest.log
timeout 900 bash tools/gpu_e2e.sh > gpurun_out/r4a/e2e.log 2>&1; grep -E "^\[compare\]|^== \[" gpurun_out/r4a/e2e.log | cut -c1-260
timeout 900 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r4a/bench.json') if l.startswith('{')][-1])
    print('randla', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['ranks_seen'])
    print('latency', {k:(v['ms_per_frame_median'] if isinstance(v,dict) else v) for k,v in d.get('latency',{}).items() if k.startswith('batch')})
    for k,w in d.get('workloads',{}).items():
        print(k, w.get('value'), {a:b for a,b in w.get('roofline',{}).items() if a in ('frac','avg_launch_ms','frac_reference_formulation','avg_launch_ms_alone','end_to_end_tflops','avg_launch_ms_alone_lane_shape','avg_launch_ms_alone_whole_batch','real_neighbours_per_query')}, w.get('error'))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r4a/bench.err').read()[-1500:])
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads > $GRAFT_REPO_ROOT/gpurun_out/r4a/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4a/prof.err
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r4a/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
find gpurun_out/r4a/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r4a/prof -name "*.db" -delete
