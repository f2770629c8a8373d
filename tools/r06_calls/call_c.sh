#!/bin/bash
# round 6, call C: first hardware contact of lfa_attn_wave_b3 (D = 64): RandLA-Net GPU tests, then the A/B (3 = all, 1 = D >= 128 only, 0 = f32)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6c
mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_randlanet.py tests/test_gpu_configs.py -q -x -k "randla or Randla or RandLA or frame_stream or tile_order or engine" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-300
( tools/r06_calls/ab_attn.sh base attn_b3_wide_only attn_f32 base attn_b3_wide_only 2>&1 ) > $O/ab.log; cat $O/ab.log
