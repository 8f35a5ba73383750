# PMC passes over the kernels of a key-frame step: one context of 128 streams, policy 1 (key-frame branch on every frame),
# profiles/tools/kf_step_stages.py; condensed by profiles/tools/kf_pmc_table.py.  One gpurun call, ~1 minute.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
K="fast_nms_kernel|blur7_kernel|orb_select_kernel|hamming_knn2_kernel|brief_kernel|ic_angle_kernel|resize_exact_kernel|ransac_kernel|pnp_refine_kernel|lk_track_kernel|pyr3_kernel|lk_border_kernel"
rm -rf /tmp/kfp; mkdir -p /tmp/kfp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-include-regex "$K" --kernel-trace --pmc $set -d /tmp/kfp/q_$n -o p --output-format csv -- python3 $R/profiles/tools/kf_step_stages.py 128 1 3 > /tmp/kfp/o_$n 2> /tmp/kfp/e_$n; echo "$n rc=$?"
done
python3 $R/profiles/tools/kf_pmc_table.py /tmp/kfp > $R/gpurun_out/${1:-r03k}_kf_kernels_pmc.json
