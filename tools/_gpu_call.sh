cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3fin8; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/pf -o kp -- python $R/bench.py --workload kpconv --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/pf -o pp -- python $R/bench.py --workload pointpillars --no-cpu-baseline > /dev/null 2>&1
cd $R
for w in kp pp; do python profiles/summarize_rocpd.py $O/pf/${w}_results.db > $O/${w}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/pf; head -5 $O/kp_kernel_stats.csv | cut -c1-120; head -4 $O/pp_kernel_stats.csv | cut -c1-120
