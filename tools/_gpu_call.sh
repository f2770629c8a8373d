cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_knn.py tests/test_gpu_randlanet.py -q 2>&1 | tail -3
echo "== knn_only: $(python tools/knn_only.py 7 2>&1 | tail -1)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4g/prof -- python $GRAFT_REPO_ROOT/tools/latency_only.py 100 > $GRAFT_REPO_ROOT/gpurun_out/r4g/lat.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r4g/prof -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-180
find gpurun_out/r4g/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r4g/prof -name "*.db" -delete
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=[d['roofline']]+d['roofline_other']
print('%.0f frames/s step %.2f ms' % (d['value'], d['ms_per_step']), ' | '.join('%s %.3f ms (alone %.3f)' % (x['kernel'][:18], x['avg_launch_ms'], x['avg_launch_ms_alone'] or 0) for x in r))
print('latency', {k:(round(v['ms_per_frame_median'],3), round(v['ms_per_frame_p95'],3)) for k,v in d['latency'].items() if k.startswith('batch')})"
done
