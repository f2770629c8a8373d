cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2o
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-workloads > $R/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads > $R/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads > $R/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/prof_kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline > $R/prof_kp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/prof_pp -- python $GRAFT_REPO_ROOT/bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline > $R/prof_pp.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $R/prof -name "*.db"); do python profiles/summarize_rocpd.py $f $R/r02_kernel_stats.csv > /dev/null; done
for f in $(find $R/prof_kp -name "*.db"); do python profiles/summarize_rocpd.py $f $R/r02_kp_kernel_stats.csv > /dev/null; done
for f in $(find $R/prof_pp -name "*.db"); do python profiles/summarize_rocpd.py $f $R/r02_pp_kernel_stats.csv > /dev/null; done
for f in $(find $R/pmc_fetch -name "*.db"); do python profiles/summarize_pmc.py $f $R/r02_pmc_fetch.csv > /dev/null; done
for f in $(find $R/pmc_write -name "*.db"); do python profiles/summarize_pmc.py $f $R/r02_pmc_write.csv > /dev/null; done
find $R -name "*.db" -size +5M -delete
tools/micro/valu_rates > $R/r02_micro_valu_rates.log 2>&1
tail -1 $R/prof_bench.log | cut -c1-600; head -8 $R/r02_kernel_stats.csv | cut -c1-150; grep -c . $R/r02_pmc_fetch.csv $R/r02_pmc_write.csv
