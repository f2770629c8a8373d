cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for rep in 1 2; do
for v in base knn_w6g2 knn_w6 knn_g2; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/variants/$v.so $LIB/libml3d_hip.so; fi
  echo "== $v $(timeout 120 python tools/knn_only.py 7 2>&1 | grep knn_only)" >> gpurun_out/r2u/abl.log
done
done
cp /tmp/base.so $LIB/libml3d_hip.so
cat gpurun_out/r2u/abl.log
