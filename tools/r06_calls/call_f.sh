#!/bin/bash
# round 6, call F: deep Linears of the RandLA forward on the bf16 pipe (A/B by K threshold), SQ counters of the bf16x3 attention kernels
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6f
mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_randlanet.py tests/test_gpu_configs.py tests/test_gpu_knn.py -q -x -k "randla or Randla or RandLA or frame_stream or tile_order or engine or knn" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-300
( tools/r06_calls/ab_attn.sh base lin_f32 lin_b3_256 lin_b3_64 base lin_f32 2>&1 ) > $O/ab.log; cat $O/ab.log | cut -c1-120
rm -rf /tmp/pmc_sq
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -f csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames-per-step 64 --no-overlap --no-cpu-baseline --no-workloads --no-latency > /tmp/pmc_sq.log 2>&1)
python tools/summarize_pmc.py /tmp/pmc_sq $O/r06_pmc_attn_sq.csv
grep -E "lfa_attn|knn_query|mlp_wave|Kernel" $O/r06_pmc_attn_sq.csv | cut -c1-330
