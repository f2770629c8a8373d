#!/bin/bash
# round 6, call S: kernel tables: pillar features alone (fill vs pfn split), KPConv bench after the bf16x3 Linears
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6s
mkdir -p $O
rm -rf /tmp/prof_pil /tmp/prof_kp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_pil -o pil -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py pillars 20 > /tmp/prof_pil.log 2>&1)
grep -v "^[EW]2026" /tmp/prof_pil.log | tail -3
cp $(find /tmp/prof_pil -name "*kernel_stats.csv" | head -1) $O/pillars_kernel_stats.csv
head -8 $O/pillars_kernel_stats.csv | cut -c1-200
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency > /tmp/prof_kp.log 2>&1)
grep -v "^[EW]2026" /tmp/prof_kp.log | tail -1 | cut -c1-300
cp $(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1) $O/kp_kernel_stats.csv
head -30 $O/kp_kernel_stats.csv | cut -c1-150
