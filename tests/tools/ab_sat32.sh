cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset HM_AMD_LIB; else export HM_AMD_LIB=$GRAFT_REPO_ROOT/build_ab/libhm_$v.so; fi
  echo "== $v"; python3 tests/tools/fp32_mode_timing.py 1000 0 2>&1 | tail -1 | sed 's/.*saturation/saturation/'; python3 tests/tools/large_grid_timing.py 512 125 8 0 32 2>&1 | tail -1 | sed 's/.*saturation/saturation/'
done
