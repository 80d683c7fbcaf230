import sys, time
sys.path.insert(0, '/root/repo')
import bench
from historymatching_amd.forward import ForwardPlan
model = bench.build_model(64, device=0)
for hist in (True, False):
    for _ in range(3):
        t0 = time.perf_counter(); plan = ForwardPlan(model, 1000, bench.DT, bench.NTIME, keep_history=hist, device=0); t1 = time.perf_counter(); plan.close(); t2 = time.perf_counter()
        print(f"keep_history={hist}: create {1e3*(t1-t0):.1f} ms, destroy {1e3*(t2-t1):.1f} ms")
