#!/bin/bash
# round 5: SQ counters of the bf16x3 convolution (3x3 64 -> 64 at 16 sweeps) alone in a process: matrix-pipe busy share, waits
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5bk
mkdir -p $O
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | cut -c1-20 | tr ' ' '_')
  rm -rf /tmp/pmc_$n
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -f csv -d /tmp/pmc_$n -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py pp 4 > /tmp/pmc_$n.log 2>&1)
  tail -3 /tmp/pmc_$n.log
  python tools/summarize_pmc.py /tmp/pmc_$n $O/pmc_$n.csv
  grep -i "bf3\|Kernel" $O/pmc_$n.csv | head -12
done
