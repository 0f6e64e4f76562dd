cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 2 3; do
cd /tmp && GSR_DBG_SCATTER=$v timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/abl_$v -o abl -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/abl_$v -name "*.db" | head -1)
echo "variant $v"; python tools/rocpd_kernel_stats.py $DB | grep -E "group_chunk|group_scatter|preprocess_kernel"
rm -rf gpurun_out/abl_$v
done
