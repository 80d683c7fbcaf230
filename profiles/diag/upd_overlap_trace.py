"""One analysis step at config 3's shape with the small fp64 chain on a second stream: run under
   rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 profiles/diag/upd_overlap_trace.py [overlap kc small]
and print the kernels' start / end times of the last step with  --report DIR."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    last = [i for i, r in enumerate(rows) if "k_center_obs" in r["Kernel_Name"] or "k_center_gram" in r["Kernel_Name"] or "k_front_chain" in r["Kernel_Name"]][-1]
    t0 = int(rows[last]["Start_Timestamp"])
    for r in rows[last:last + 8]:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us  queue {r.get('Queue_Id', '?'):>3}  "
              f"{r['Kernel_Name'][:60]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))} lds {r.get('LDS_Block_Size', '?')} vgpr {r.get('VGPR_Count', '?')} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size', '?'))}")
    sys.exit(0)

import numpy as np
import scipy.linalg as sla

from historymatching_amd import _lib
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan

ov, kc, sm = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 32, 1)))
ff = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
R12 = obs_error_model(40, 4)[1]
p = UpdatePlan(N, N, M, n_obs, dtype=32)
p.set_option("overlap", ov)
p.set_option("gxt_chunk", kc)
p.set_option("small_inverse", sm)
p.set_option("fused_front", ff)
p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
p.run_local()
for _ in range(5):
    _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
print(p.sync()["ms_update"] / 5)
