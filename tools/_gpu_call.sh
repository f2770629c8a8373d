cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3ag; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && ML3D_KNN_PHASES=1 rocprofv3 --kernel-trace --stats -d $R/$O/pf -o k -- python $R/tools/knn_only.py 3 > /dev/null 2>&1
cd $R; python profiles/summarize_rocpd.py $O/pf/k_results.db 2>/dev/null | grep -E "knn_query|Name" | cut -c1-200
