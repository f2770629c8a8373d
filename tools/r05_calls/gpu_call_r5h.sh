#!/bin/bash
# round 5: k-NN instruction trim (in-place insertion chain, 32-bit offsets, xy-packed distance) -- parity, launch time, headline
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_randlanet.py tests/test_gpu_api.py -x -q 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 300 python tools/knn_only.py 5 2>&1 | tail -3 ) > $O/knn_only.log; cat $O/knn_only.log
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-workloads --no-latency --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%.1f frames/s step %.3f ms | knn in-region %.3f alone %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['avg_launch_ms_alone']))"
done > $O/bench.log 2>&1
cat $O/bench.log
rm -rf /tmp/pmc_knn
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES -f csv -d /tmp/pmc_knn -- python $GRAFT_REPO_ROOT/tools/knn_only.py 3 > /tmp/pmc_knn.log 2>&1)
python tools/summarize_pmc.py /tmp/pmc_knn $O/r05b_pmc_knn_sq.csv
grep knn_query $O/r05b_pmc_knn_sq.csv | cut -c1-160
