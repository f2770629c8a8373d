cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3ae; mkdir -p $O
export TMPDIR=/tmp
ML3D_BENCH_PROFILE=1 timeout 600 python bench.py --no-workloads --no-cpu-baseline --steps 5 --warmup 2 > $O/b.json 2> $O/prof.txt
grep -A45 "latency profile, batch 1" $O/prof.txt | cut -c1-180 | head -60
