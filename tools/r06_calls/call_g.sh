#!/bin/bash
# round 6, call G: mlp_chain_b3 (per-point chains on the bf16 pipe, activations in registers): RandLA tests + A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6g
mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_randlanet.py tests/test_gpu_configs.py tests/test_gpu_api.py tests/test_gpu_pipelines.py -q -x -k "randla or Randla or RandLA or frame_stream or tile_order or engine" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-300
( tools/r06_calls/ab_attn.sh base chain_f32 base chain_f32 2>&1 ) > $O/ab.log; cat $O/ab.log | cut -c1-200
( timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-workloads --no-latency --no-overlap --breakdown 2>$O/b1.err | tail -1 ) > $O/bench_seq.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6g/bench_seq.json').read())
b=d.get('breakdown_ms',{})
print('sequential', d['value'], ' '.join('%s=%.3f'%(k,b[k]) for k in ('fwd:1200','fwd:13','fwd:22','fwd:30','fwd:21','fwd:29','fwd:1001','fwd:1100','fwd:1101','fwd:1102','fwd:11','fwd:3','knn:0','knn:100') if k in b))
PY
