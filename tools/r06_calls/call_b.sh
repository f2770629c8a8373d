#!/bin/bash
# round 6, call B: first hardware contact of lfa_attn_b3 (bf16x3 attention, D = 128 / 256): the RandLA-Net GPU tests, then the A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6b
mkdir -p $O
( timeout 120 tools/micro/lds_corun 2>&1 | tail -30 ) > $O/lds_corun.log; head -9 $O/lds_corun.log | cut -c1-200; tail -4 $O/lds_corun.log | cut -c1-200
( timeout 500 python -m pytest tests/test_gpu_randlanet.py tests/test_gpu_configs.py -q -x -k "randla or Randla or RandLA or frame_stream or tile_order or engine" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-300
( tools/r06_calls/ab_attn.sh base attn_f32 b3_tp44 b3_tp22 base attn_f32 2>&1 ) > $O/ab.log; cat $O/ab.log
