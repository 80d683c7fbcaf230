import sys, time
sys.path.insert(0,'/root/repo')
from historymatching_amd.forward import ForwardPlan
from tests.helpers import make_models, perms
_, gm = make_models(20, 20)
N=100
x = perms(20, 20, N, seed=1)
for tp, ts in ((0,0),(60,0),(120,0),(240,64),(240,128),(60,64),(60,128),(20,64),(40,128)):
    plan = ForwardPlan(gm, N, 0.025, 40, keep_history=True)
    if tp: plan.set_debug("threads_pressure", tp)
    if ts: plan.set_debug("threads_saturation", ts)
    plan.set_inputs(x, None, transformed=False)
    plan.run(); plan.sync()
    plan.set_inputs(x, None, transformed=False)
    t0=time.perf_counter(); plan.run(); st=plan.sync(); t=time.perf_counter()-t0
    print(f"threads pressure {tp or 240} saturation {ts or 256}: {t*1e3:.2f} ms per pass; pressure {st['ms_pressure']/st['n_pressure_launches']*1e3:.1f} us, saturation {st['ms_saturation']/st['n_saturation_launches']*1e3:.1f} us per launch")
    plan.close()
