cd /tmp && export TMPDIR=/tmp
R=/root/repo
B="python3 $R/bench.py --contexts 1 --batch 256 --steps 4 --warmup 1 --ingest-steps 0 --extra-steps 0 --single-steps 0 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats -d /tmp/s1 -o s --output-format csv -- $B > $R/gpurun_out/r03k_1x256_under_rocprof.json 2> /tmp/e1; echo stats rc=$?
grep -v "at::native\|rocclr\|anonymous" /tmp/s1/s_kernel_stats.csv > $R/gpurun_out/r03k_kernel_stats_1x256.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-include-regex "lk_track_kernel|pyr3_kernel" --kernel-trace --pmc $set -d /tmp/p_$n -o p --output-format csv -- $B > /dev/null 2> /tmp/e_$n; echo "$n rc=$?"
  f=$(find /tmp/p_$n -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/tools/condense_pmc.py $f > $R/gpurun_out/r03k_pmc_$n.json
done
rocprofv3 --kernel-trace --stats -d /tmp/s2 -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pmc > $R/gpurun_out/r03k_default_under_rocprof.json 2> /tmp/e2; echo default-stats rc=$?
grep -v "at::native\|rocclr\|anonymous" /tmp/s2/s_kernel_stats.csv > $R/gpurun_out/r03k_kernel_stats_default_bench.csv
