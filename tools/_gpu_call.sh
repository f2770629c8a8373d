cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4p/bench.json').readline())
print(d['value'], d['unit'], d['ms_per_step'])
for w in ('kpconv','pointpillars'):
    x=d.get('workloads',{}).get(w)
    if x: print(w, x['value'], x['ms_per_step'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in x['roofline'].items() if k in ('frac','frac_alone','avg_launch_ms','avg_launch_ms_alone','end_to_end_tflops')})
l=d.get('latency'); print('latency', l['batch_1']['ms_per_frame_median'], l['batch_4']['ms_per_frame_median'])
PY
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/profiles; mkdir -p $OUT
rm -rf /tmp/prof_kp; (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency > /tmp/prof_kp.log 2>&1)
cp $(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1) $OUT/r04_kp_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do n=$(echo $c | tr A-Z a-z | sed 's/_size//'); rm -rf /tmp/pmc_kp_$n; (cd /tmp && rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_kp_$n -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py kp 4 > /dev/null 2>&1); python tools/summarize_pmc.py /tmp/pmc_kp_$n $OUT/r04_pmc_kp_$n.csv; done
python tools/make_traffic.py --op kpconv_block_32_32 $OUT/r04_pmc_kp_fetch.csv $OUT/r04_pmc_kp_write.csv | tail -3
cp profiles/traffic.json $OUT/traffic.json
