# VALU instructions / HBM bytes per tracked point of lk_track_kernel for build variants of the library (tools/build_variant.sh):
#   bash profiles/tools/pmc_variants.sh "" /root/repo/build/libmvo_X.so ...      ("" = the product build)
cd /root/repo
for v in "$@"; do
  MVO_LIB=$v timeout -k 10 200 python3 -c "
import bench, types, json
a = types.SimpleNamespace(width=1280, height=720, nfeatures=2000, max_points=None)
o = bench.run_pmc(a)
print('variant [$v]', json.dumps({k: o.get(k) for k in ('valu_instructions_per_point','hbm_read_bytes_per_point')}), [p.get('kernel_ms_avg_under_pmc') for p in o.get('passes', [])])
" || exit 1
done
