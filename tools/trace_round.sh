# kernel trace of the default bench condensed on the box (the trace itself is too large to bring back): gpurun_out/TAG_step_timeline*.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
T=${1:-r03k}
rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python3 $R/bench.py --steps 22 --warmup 3 --no-cpu-baseline --no-pmc --ingest-steps 0 --single-steps 0 --extra-steps 0 > $R/gpurun_out/${T}_bench_under_trace.json 2> /tmp/tr_err; echo rc=$?
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
grep -v "at::native\|rocclr\|anonymous" $f > /tmp/tr/ours.csv
python3 $R/profiles/tools/step_timeline.py /tmp/tr/ours.csv > $R/gpurun_out/${T}_step_timeline.txt
python3 $R/profiles/tools/step_timeline.py /tmp/tr/ours.csv --steps > $R/gpurun_out/${T}_step_timeline_steps.txt
