#!/bin/bash
# round 6, call D: the whole GPU suite on the bf16x3 attention kernels, the kernel table of the RandLA step, the default bench line
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6d
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
rm -rf /tmp/prof_randla
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_randla -o randla -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency > /tmp/prof_randla.log 2>&1)
cp $(find /tmp/prof_randla -name "*kernel_stats.csv" | head -1) $O/r06_randla_kernel_stats.csv
head -28 $O/r06_randla_kernel_stats.csv | cut -c1-150
( timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-600 $O/bench.json
