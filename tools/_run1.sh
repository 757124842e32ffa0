cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03_rowmajor
mkdir -p $OUT
python3 bench.py --order row_major --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 bench.py --no-cpu-baseline --order row_major --steps 20 --warmup 5 > $OUT/bench_stats.log 2>&1
grep -h '^{"metric"' $OUT/bench_plain.log | cut -c1-1500
find $OUT/stats -name "*kernel_stats.csv" | head -2
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -8 "$f"
