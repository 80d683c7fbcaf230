#!/bin/bash
# pressure tests on all three grids, then the config-4 / config-5 shard timings (pressure ms per launch)
timeout 900 python -m pytest tests/test_forward_gpu.py tests/test_configs_gpu.py -x -q -k "nested or assembly or direct_solver or ill_conditioned or hand_over or config5_grid or config4 or embedded" 2>&1 | tail -3
python tests/tools/large_grid_timing.py 256 512 39 2>&1 | tail -1
python tests/tools/large_grid_timing.py 512 125 39 0 32 2>&1 | tail -1
