cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3aa; mkdir -p $O
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base q_w5 q_w4 noq q_w5; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  echo "$v: $(timeout 120 python tools/knn_only.py 7 2>&1 | tail -1)"
  timeout 200 python bench.py --no-cpu-baseline --no-workloads --no-latency > $O/rl_$v.json 2> $O/rl_$v.err
  echo "$v: $(python -c "import json; d=json.load(open('$O/rl_$v.json')); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],3), r['kernel'][:20], round(r['avg_launch_ms'],3), r.get('avg_launch_ms_alone'))" 2>&1 | tail -1)"
done
cp /tmp/base.so $LIB/libml3d_hip.so
