#!/bin/bash
# round 5: polygon lists of the rotated IoU in registers instead of LDS: the decode tail under a co-running bf16x3 forward, the two-lane
# pipeline against one lane, the detection / NMS / IoU GPU tests, nmsb_mask's launch time
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zg
mkdir -p $O
( timeout 100 python tools/r05_calls/diag_decode_variants.py 2>&1 | tail -12 ) > $O/new_regs.log
( timeout 100 python tools/r05_calls/diag_two_lane2.py 2>&1 | tail -12 ) > $O/new_two_lane.log
( timeout 300 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_prims.py tests/test_gpu_api.py -q -x -k "bboxes or nms or iou or two_lane or both_conv or 16_sweeps or detect" 2>&1 | tail -6 ) > $O/pytest.log
for f in new_regs new_two_lane pytest; do echo "== $f"; cut -c1-500 $O/$f.log; done
rm -rf /tmp/kt; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o pp -- python $GRAFT_REPO_ROOT/bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline --no-latency > /tmp/kt.log 2>&1)
tail -1 /tmp/kt.log | cut -c1-200
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_pp_kernel_stats.csv
grep "nmsb\|iou" $O/r05_pp_kernel_stats.csv | cut -c1-200
