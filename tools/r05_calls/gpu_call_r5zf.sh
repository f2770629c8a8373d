#!/bin/bash
# round 5: decode tail under a co-running bf16x3 forward -- hipMemsetAsync (old library) against the fill kernel (new), torch.empty against zeroed workspaces
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zf
mkdir -p $O
OLD=open3d-ml_amd/ml3d/lib/ab/memset_old.so
( ML3D_DIAG_LIB=$OLD timeout 100 python tools/r05_calls/diag_decode_variants.py 2>&1 | tail -12 ) > $O/old_empty.log
( ML3D_DIAG_LIB=$OLD ML3D_DIAG_WS=zeros timeout 100 python tools/r05_calls/diag_decode_variants.py 2>&1 | tail -12 ) > $O/old_zeros.log
( timeout 100 python tools/r05_calls/diag_decode_variants.py 2>&1 | tail -12 ) > $O/new_empty.log
( timeout 100 python tools/r05_calls/diag_two_lane2.py 2>&1 | tail -12 ) > $O/new_two_lane.log
for f in old_empty old_zeros new_empty new_two_lane; do echo "== $f"; cut -c1-700 $O/$f.log; done
