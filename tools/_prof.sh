set -e
O=gpurun_out/r02z; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.txt 2> $O/bench_default.err
echo bench done
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/bench_under_rocprof.txt 2>&1
cd $R
cp $(find $O/stats -name '*kernel_stats.csv') $O/kernel_stats_default_bench.csv
python3 profiles/tools/kernel_durations.py $(find $O/stats -name '*kernel_trace.csv') 'lk_track_kernel' | head -1 > $O/lk_launches_default.txt
find $O -name '*.csv' -size +500k -delete
cat $O/lk_launches_default.txt
