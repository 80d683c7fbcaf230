cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in base nop1 nored nop1red; do
  if [ $v = base ]; then unset HM_AMD_LIB; else export HM_AMD_LIB=$GRAFT_REPO_ROOT/build_ab/libhm_$v.so; fi
  rm -rf /tmp/ib; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ib -o ib -- python3 tests/tools/ies_step_profile.py 1000 1 160 3 > /dev/null 2>&1
  echo "== $v"; python3 profiles/tools/print_stats.py /tmp/ib | grep -E "k_ib_panel|k_ib_backsolve|k_ib_swap"
done
