# peak rate of scattered one-byte gathers (distinct 128-byte lines) on this device: the calibration probe under a kernel trace
export TMPDIR=/tmp
OUT=gpurun_out/r04_gather_rate
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- $GRAFT_REPO_ROOT/profiles/probes/calib_gather 8 30 > $GRAFT_REPO_ROOT/$OUT/run.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $OUT/run.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
