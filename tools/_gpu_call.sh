cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/profiles
mkdir -p $OUT
pmc() {
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -f csv -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  python tools/summarize_pmc.py /tmp/pmc_$name $OUT/r04_pmc_$name.csv
}
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
C2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
pmc kp_sq1 "$C1" python $GRAFT_REPO_ROOT/tools/roofline_ops.py kp 4
pmc kp_sq2 "$C2" python $GRAFT_REPO_ROOT/tools/roofline_ops.py kp 4
STEP="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads --no-latency"
pmc forward_sq1 "$C1" $STEP
grep "kp_agg_gemm32" $OUT/r04_pmc_kp_sq1.csv $OUT/r04_pmc_kp_sq2.csv | cut -d, -f2- | cut -c1-120
grep "lfa_attn_mfma16\|head_fc0" $OUT/r04_pmc_forward_sq1.csv | cut -c1-200
