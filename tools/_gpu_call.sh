cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pointpillars.py tests/test_gpu_prims.py tests/test_gpu_knn.py -q 2>&1 | tail -3 > gpurun_out/r2t/ab.log
for rep in 1 2; do
timeout 300 python bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('kpconv spheres/s %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))" >> gpurun_out/r2t/ab.log
done
timeout 300 python bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('pointpillars frames/s %.0f' % d['value'], 'conv 64->64: %.3f ms %.1f TF frac %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['roofline']['frac']))" >> gpurun_out/r2t/ab.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2t/prof_kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/r2t/prof_kp -name "*.db"); do python profiles/summarize_rocpd.py $f gpurun_out/r2t/kp_kernel_stats.csv > /dev/null; done
find gpurun_out/r2t -name "*.db" -size +5M -delete
cat gpurun_out/r2t/ab.log; grep -E "grid_bbox|grid_occupancy|grid_hist|grid_scatter" gpurun_out/r2t/kp_kernel_stats.csv | cut -c1-160
