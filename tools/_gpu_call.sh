cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3v; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pipelines.py -x -q 2>&1 | tail -2
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --workload kpconv --no-cpu-baseline > $O/kp_$name.json 2> $O/kp_$name.err
  echo "kpconv $name: $(python -c "import json; d=json.load(open('$O/kp_$name.json')); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],3), 'block ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))" 2>&1 | tail -1)"
}
run fuse ML3D_KP_FUSE_SHORTCUT=1
run nofuse ML3D_KP_FUSE_SHORTCUT=0
run fuse_b ML3D_KP_FUSE_SHORTCUT=1
run nofuse_b ML3D_KP_FUSE_SHORTCUT=0
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python profiles/summarize_rocpd.py $O/prof/kp_results.db 2>/dev/null | head -12 | cut -c1-170
