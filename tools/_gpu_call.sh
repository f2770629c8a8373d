cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3ar; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kpconv.py -x -q 2>&1 | tail -2
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --workload kpconv --no-cpu-baseline --steps 30 --warmup 8 > $O/kp_$name.json 2> $O/kp_$name.err
  echo "kpconv $name: $(python -c "import json; d=json.load(open('$O/kp_$name.json')); r=d['roofline']; print(round(d['value'],1), round(d['step_ms_median'],3), 'block ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))" 2>&1 | tail -1)"
}
run n32 ML3D_GEMM_N32=1
run tile ML3D_GEMM_N32=0
run n32_b ML3D_GEMM_N32=1
run tile_b ML3D_GEMM_N32=0
