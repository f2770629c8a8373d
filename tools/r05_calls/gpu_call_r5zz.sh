#!/bin/bash
# round 5, closing call of the re-entry session: full GPU suite + smoke at HEAD, PMC traffic of SECOND's 3x3 convolution on the bf16x3 kernel,
# the default bench line, the training-step A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zz
mkdir -p $O
rm -f gpurun_out/parity_per_yaml.jsonl
( timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
python tools/parity_table.py gpurun_out/parity_per_yaml.jsonl > $O/parity_per_yaml.md 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $O/smoke.log; cat $O/smoke.log
pmc() {     # out csv name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --pmc $ctr -f csv -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  python tools/summarize_pmc.py /tmp/pmc_$name $O/r05_pmc_$name.csv
}
pmc pp_fetch FETCH_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py pp 4
pmc pp_write WRITE_SIZE python $GRAFT_REPO_ROOT/tools/roofline_ops.py pp 4
python tools/make_traffic.py --op pp_conv3x3_64 $O/r05_pmc_pp_fetch.csv $O/r05_pmc_pp_write.csv 2>&1 | tail -2
cp profiles/traffic.json $O/traffic.json
( timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-260 $O/bench.json
( timeout 150 python tools/train_step_ab.py randlanet 4 torch,hip,torch,hip 2>&1 | grep -v "return float" | tail -5 ) > $O/train_ab_randlanet.log; cat $O/train_ab_randlanet.log
