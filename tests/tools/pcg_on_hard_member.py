"""What a single-workgroup Jacobi-CG (press_variant 9) and the two-level CG (15) cost on ONE member at 256 x 256 -- the candidates for a
device-side hand-over of members the direct solver cannot solve: an ordinary member and the ill-conditioned one of config 4's prior
(K = 0.1 ... 1.2e9).   python tests/tools/pcg_on_hard_member.py [steps=3]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from tests.helpers import make_models, perms  # noqa: E402
from tests.test_forward_gpu import _config4_member_2086  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 256
_, gm = make_models(n, n)
for name, x in (("ordinary member", perms(n, n, 1, seed=5)), ("member 2086 of config 4 (K up to 1.2e9)", _config4_member_2086(n))):
    for variant in (9, 15, 12):
        plan = ForwardPlan(gm, 1, 0.025, steps, keep_history=False)
        plan.set_variant(variant, 0)
        plan.set_inputs(x, None, transformed=False)
        t0 = time.perf_counter()
        plan.run()
        st = plan.sync()
        wall = time.perf_counter() - t0
        _, _, status = plan.outputs(want_wsats=False)
        print(f"{name}, press_variant {variant}: pressure {st['ms_pressure'] / st['n_pressure_launches']:.1f} ms per step, mean CG iterations {st['mean_n_cg']:.0f}, "
              f"status {int(status[0])}, wall {wall:.2f} s", flush=True)
        plan.close()
