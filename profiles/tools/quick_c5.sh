#!/bin/bash
mkdir -p gpurun_out
HM_AMD_LIB=build_ab/libhm_r5base.so python profiles/diag/c5_smax.py 125 base 2>&1 | tail -4
python profiles/diag/c5_smax.py 125 new 2>&1 | tail -4
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/c5_P1_base.npy"); b=np.load("gpurun_out/c5_P1_new.npy")
print("first-step pressure, 4 members: max |new - base| / max |P| =", np.abs(a-b).max()/np.abs(a).max(), " identical:", np.array_equal(a,b))
PY
