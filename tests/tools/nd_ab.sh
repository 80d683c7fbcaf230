cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/ks_$v
  HM_AMD_LIB=build_ab/libhm_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o ks -- python3 tests/tools/nd_time.py 12 1000 1 2>&1 | grep "variant 12" | sed "s/^/$v: /"
  python3 profiles/tools/print_stats.py /tmp/ks_$v/ks_kernel_stats.csv 2>&1 | grep -E "k_nd_top|k_nd_sub|k_nd_wave|k_nd_solve\(|k_nd_leaf\(" | sed "s/^/$v: /" | cut -c1-40,100-140
done
