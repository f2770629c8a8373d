cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kpconv.py -x -q 2>&1 | tail -6
timeout 200 python bench.py --workload kpconv --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('kpconv', round(d['value'],1), d['step_ms_median'], d['roofline']['frac'])"
