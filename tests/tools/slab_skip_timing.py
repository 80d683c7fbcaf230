"""Per-time-step device time of the float32 slab sweep at 512 x 512 with the ensemble in one launch (dry slabs leave at once) and in rounds of
co-resident teams (hm_fwd_set_debug "team_rounds"):   python tests/tools/slab_skip_timing.py [N=125] [steps=40] [grid=512]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from tests.helpers import make_models, perms  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
_, gm = make_models(n, n, dtype=32)
x = perms(n, n, N, seed=3)
for rounds in (1, 0):
    plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False)
    plan.set_debug("team_rounds", rounds)
    plan.set_inputs(x, None, transformed=False)
    line = []
    t0 = time.perf_counter()
    for k in range(steps):
        t1 = time.perf_counter()
        plan.run(k, 1)
        st = plan.sync()
        line.append((time.perf_counter() - t1) * 1e3)
    _, _, status = plan.outputs(want_wsats=False)
    print(f"{'rounds of co-resident teams' if rounds else 'one launch, dry slabs leave'}: wall ms per step (pressure + sweep): " + " ".join(f"{v:.0f}" for v in line) +
          f"; wall {time.perf_counter() - t0:.2f} s; status ok {not status.any()}; member-steps redone by the tiled sweep: {st['team_retries']}, by the slab sweep's redo launch: {st['slab_redos']}", flush=True)
    plan.close()
