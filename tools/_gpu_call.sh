cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_training.py -q > gpurun_out/r4h/train.log 2>&1; tail -8 gpurun_out/r4h/train.log
grep -n "Fatal\|fault\|Error\|error\|File \"/.*repo" gpurun_out/r4h/train.log | head -30
