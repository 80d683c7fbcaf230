# timing of sat256s.hip variants (build_ab/libhm_<name>.so from profiles/diag/build_src_ab.sh): bash tests/tools/ab_slab.sh name...
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset HM_AMD_LIB; else export HM_AMD_LIB=$GRAFT_REPO_ROOT/build_ab/libhm_$v.so; fi
  echo "== $v"; python3 tests/tools/large_grid_timing.py 256 512 3 0 64 0 2>&1 | tail -1 | sed 's/.*saturation/saturation/'; python3 tests/tools/large_grid_timing.py 256 512 16 0 64 0 2>&1 | tail -1 | sed 's/.*saturation/saturation/'
done
