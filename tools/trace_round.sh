cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python3 $R/bench.py --steps 22 --warmup 3 --no-cpu-baseline --no-pmc --ingest-steps 0 --single-steps 0 --extra-steps 0 > $R/gpurun_out/r03ay_bench_under_trace.json 2> /tmp/tr_err; echo rc=$?
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
grep -v "at::native\|rocclr\|anonymous" $f > /tmp/tr/ours.csv
python3 $R/profiles/tools/step_timeline.py /tmp/tr/ours.csv > $R/gpurun_out/r03ay_step_timeline.txt
python3 $R/profiles/tools/step_timeline.py /tmp/tr/ours.csv --steps > $R/gpurun_out/r03ay_step_timeline_steps.txt
