# timing of sat128r.hip variants (build_ab/libhm_<name>.so from profiles/diag/build_src_ab.sh): bash tests/tools/ab_sat128r.sh name...
cd $GRAFT_REPO_ROOT
for v in base "$@" base; do
  if [ $v = base ]; then unset HM_AMD_LIB; else export HM_AMD_LIB=$GRAFT_REPO_ROOT/build_ab/libhm_$v.so; fi
  echo "== $v"; python3 tests/tools/sat_time_only.py 1000 40 0 2>&1 | tail -1
done
