#!/bin/bash
# GPU box: the round's evidence under gpurun_out/<tag>/ -- kernel tables of the three bench commands, FETCH_SIZE / WRITE_SIZE
# counter passes (separate passes, as MI355X_MICROARCH.md prescribes) of the RandLA step and of the KPConv / PointPillars
# roofline ops run alone, the SQ counters of the k-NN launch, and profiles/traffic.json rebuilt from them.
# usage: bash tools/gpu_round_profiles.sh r04   (then copy gpurun_out/r04/profiles/* into profiles/)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/profiles
mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$name -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/prof_$name.log 2>&1)
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
  tail -1 /tmp/prof_$name.log | cut -c1-200
}
pmc() {     # out csv name, counters (quoted), command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -f csv -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  python tools/summarize_pmc.py /tmp/pmc_$name $OUT/${TAG}_pmc_$name.csv
}
stats randla --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency
stats kp --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency
stats pp --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline --no-latency
STEP="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames-per-step 64 --no-overlap --no-cpu-baseline --no-workloads --no-latency"
pmc fetch FETCH_SIZE $STEP
pmc write WRITE_SIZE $STEP
pmc kp_fetch FETCH_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py kp 4
pmc kp_write WRITE_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py kp 4
pmc pp_fetch FETCH_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py pp 4
pmc pp_write WRITE_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py pp 4
prim() {    # traffic.json key, roofline_ops mode: FETCH / WRITE passes of one primitive alone -> its traffic entry
  local key=$1 mode=$2
  pmc ${mode}_fetch FETCH_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py $mode 4
  pmc ${mode}_write WRITE_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py $mode 4
  PRIM_ARGS="$PRIM_ARGS $key:$mode"
}
PRIM_ARGS=""
prim kp_radius_dense radius
prim kp_subsample subsample
prim pp_voxelize voxelize
prim pp_pillar_features pillars
pmc knn_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES" python $GRAFT_REPO_ROOT/tools/knn_only.py 3
cp profiles/traffic.json /tmp/traffic_before.json
python tools/make_traffic.py $OUT/${TAG}_pmc_fetch.csv $OUT/${TAG}_pmc_write.csv 64
python tools/make_traffic.py --op kpconv_block_32_32 $OUT/${TAG}_pmc_kp_fetch.csv $OUT/${TAG}_pmc_kp_write.csv
python tools/make_traffic.py --op pp_conv3x3_64 $OUT/${TAG}_pmc_pp_fetch.csv $OUT/${TAG}_pmc_pp_write.csv
for kv in $PRIM_ARGS; do
  python tools/make_traffic.py --prim ${kv%%:*} $OUT/${TAG}_pmc_${kv##*:}_fetch.csv $OUT/${TAG}_pmc_${kv##*:}_write.csv 5
done
python tools/make_traffic.py --sq "knn_query_multi<16, true>" $OUT/${TAG}_pmc_knn_sq.csv
cp profiles/traffic.json $OUT/traffic.json
ls -la $OUT
