#!/bin/bash
# quick check of the saturation sweeps on one box: bit-exactness tests, variants against each other over a whole run, a short bench line
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_forward_gpu.py -x -q -k "bitexact or adversarial or multi_tile or saturation" > gpurun_out/r05/t2.txt 2>&1; tail -3 gpurun_out/r05/t2.txt
python tests/tools/sat_check.py 1000 5,1,0 6 40 > gpurun_out/r05/sat_check.txt 2>&1; tail -6 gpurun_out/r05/sat_check.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams --no-host-call --no-config5 > gpurun_out/r05/bench_quick.json 2>gpurun_out/r05/bench_quick.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r05/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
tail -3 gpurun_out/r05/bench_quick.err
