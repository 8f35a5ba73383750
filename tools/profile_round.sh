# The round's profile set in one gpurun call: `bash tools/profile_round.sh TAG` writes gpurun_out/TAG_* (copy what is judged into profiles/).
cd /tmp && export TMPDIR=/tmp
R=/root/repo
T=${1:-r03k}
B="python3 $R/bench.py --contexts 1 --batch 256 --steps 4 --warmup 1 --ingest-steps 0 --extra-steps 0 --single-steps 0 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats -d /tmp/s1 -o s --output-format csv -- $B > $R/gpurun_out/${T}_1x256_under_rocprof.json 2> /tmp/e1; echo stats rc=$?
grep -v "at::native\|rocclr\|anonymous" /tmp/s1/s_kernel_stats.csv > $R/gpurun_out/${T}_kernel_stats_1x256.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-include-regex "lk_track_kernel|pyr3_kernel|lk_border_kernel" --kernel-trace --pmc $set -d /tmp/p_$n -o p --output-format csv -- $B > /dev/null 2> /tmp/e_$n; echo "$n rc=$?"
  f=$(find /tmp/p_$n -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/tools/condense_pmc.py $f > $R/gpurun_out/${T}_pmc_$n.json
done
rocprofv3 --kernel-trace --stats -d /tmp/s2 -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pmc > $R/gpurun_out/${T}_default_under_rocprof.json 2> /tmp/e2; echo default-stats rc=$?
grep -v "at::native\|rocclr\|anonymous" /tmp/s2/s_kernel_stats.csv > $R/gpurun_out/${T}_kernel_stats_default_bench.csv
# one context of 512 streams alone: stage times of a key-frame step (policy 1) and of an LK + PnP step (policy 2)
python3 $R/profiles/tools/kf_step_stages.py 512 1 6 > $R/gpurun_out/${T}_stages_alone_keyframe_step.json 2> /tmp/e3; echo kf-stages rc=$?
python3 $R/profiles/tools/kf_step_stages.py 512 2 6 > $R/gpurun_out/${T}_stages_alone_normal_step.json 2> /tmp/e4; echo normal-stages rc=$?
bash $R/tools/kf_pmc.sh $T
bash $R/tools/trace_round.sh $T
