#!/bin/bash
# the default bench line only (what the driver runs), twice
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5y
mkdir -p $O
for i in 1 2; do
  ( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_$i.err | tail -1 ) > $O/bench_$i.json
  python -c "
import json; d=json.load(open('$O/bench_$i.json'))
print('headline %.1f | e2e %.3f | kpconv %.1f (e2e %.3f) | pp %.1f (e2e %.3f) | latency %.3f / %.3f' % (d['value'], d['end_to_end']['frac_of_f32_mfma_peak'], d['workloads']['kpconv']['value'], d['workloads']['kpconv']['roofline']['end_to_end_frac'], d['workloads']['pointpillars']['value'], d['workloads']['pointpillars']['roofline']['end_to_end_frac'], d['latency']['batch_1']['ms_per_frame_median'], d['latency']['batch_4']['ms_per_frame_median']))"
done
