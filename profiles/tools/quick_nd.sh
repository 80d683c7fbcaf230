#!/bin/bash
# quick check of the nested-dissection pressure step on one box: cycle stamps (build_prof/), the pressure tests, a short bench line
mkdir -p gpurun_out/r05b
HM_AMD_LIB=build_prof/libhm_ndprof.so python profiles/diag/nd_prof.py 1000 > gpurun_out/r05b/nd_prof_${1:-x}.txt 2>&1
head -16 gpurun_out/r05b/nd_prof_${1:-x}.txt
timeout 900 python -m pytest tests/test_forward_gpu.py -x -q -k "nested or assembly or direct_solver or ill_conditioned or hand_over" > gpurun_out/r05b/t_press.txt 2>&1; tail -3 gpurun_out/r05b/t_press.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams --no-host-call --no-config5 > gpurun_out/r05b/bench_quick_${1:-x}.json 2>gpurun_out/r05b/bench_quick.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r05b/bench_quick_${1:-x}.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
tail -3 gpurun_out/r05b/bench_quick.err
