#!/bin/bash
# the default line of the final build, the driver's round-end command, and the N = 2 launch path rehearsed on one GPU (both ranks on device 0)
mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "default rc $?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver.json 2> gpurun_out/final/bench_driver.err; echo "driver rc $?"
HM_BENCH_ALL_ON_DEVICE0=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-config5 --no-esmda --no-host-call > gpurun_out/final/bench_2ranks_one_gpu.json 2> gpurun_out/final/bench_2ranks_one_gpu.err; echo "2 ranks rc $?"
python3 - <<'PY'
import json
for n in ("bench_default", "bench_driver", "bench_2ranks_one_gpu"):
    try:
        d = json.loads(open(f"gpurun_out/final/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["n_gpus"], d["blocks"]["n"], d["blocks"].get("one_block", {}).get("value"), d["roofline"]["frac"], d["roofline"]["pressure"]["fp64_mfma_frac"], d["config"]["results_finite_and_status_ok"])
    except Exception as e:
        print(n, "unreadable:", e)
PY
for e in gpurun_out/final/*.err; do tail -n 2 "$e"; done
