#!/bin/bash
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt && mkdir -p /tmp/kt
cd $GRAFT_REPO_ROOT
for l in t4 new; do
  if [ $l = t4 ]; then export HM_AMD_LIB=build_ab/libhm_t4.so; else unset HM_AMD_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt/$l -o ks -- python3 profiles/diag/nd_time.py 1000 10 > /tmp/kt/$l.out 2> /tmp/kt/$l.err
  echo "== $l"; tail -2 /tmp/kt/$l.out
  f=$(find /tmp/kt/$l -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith(("void (anonymous namespace)::k_nd", "(anonymous namespace)::k_nd")):
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
done
