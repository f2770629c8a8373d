cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base knn_nopf; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/variants/$v.so $LIB/libml3d_hip.so; fi
  echo "== $v" >> gpurun_out/r2j/abl.log
  timeout 120 python tools/knn_only.py 5 2>&1 | grep knn_only >> gpurun_out/r2j/abl.log
done
cp /tmp/base.so $LIB/libml3d_hip.so
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/r2j/abl.log
timeout 600 python bench.py > gpurun_out/r2j/bench.log 2>&1
cat gpurun_out/r2j/abl.log; tail -1 gpurun_out/r2j/bench.log
