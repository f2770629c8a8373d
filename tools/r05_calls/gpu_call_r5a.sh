#!/bin/bash
# round 5, first GPU call: the full GPU suite, the default bench line, the decode-tail A/B, the KPConv host
# timeline, then the round's profile set (kernel tables, PMC traffic of the step / the ops / the four primitives, k-NN SQ counters)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
rm -f gpurun_out/parity_per_yaml.jsonl
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log
python tools/parity_table.py gpurun_out/parity_per_yaml.jsonl > $O/parity_per_yaml.md 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-400 $O/bench.json
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/nms_old.so $LIB/libml3d_hip.so; fi
  echo "decode tail $v: $(timeout 300 python bench.py --workload pointpillars --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f frames/s, step median %.3f p95 %.3f ms, single sweep %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95'], d['latency_single_sweep_ms']['median']))")"
done > $O/decode_tail_ab.log 2>&1
cp /tmp/new.so $LIB/libml3d_hip.so
cat $O/decode_tail_ab.log
( timeout 300 python tools/kp_host_profile.py 20 2>&1 | head -60 ) > $O/kp_host_profile.log
head -3 $O/kp_host_profile.log
timeout 1500 bash tools/gpu_round_profiles.sh r05 > $O/profiles.log 2>&1
tail -30 $O/profiles.log
