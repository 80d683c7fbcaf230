"""One IES iterate at config 3's size (N_e = 1000, 128 x 128, n_obs = 160): the forward pass on the GPU beside the host's subspace
algebra in its two forms (update.ies_step: "gram" = LU solve + 160 x 160 Cholesky, "svd" = the reference's pinv(W) + SVD(Y0),
HistoryMatch.py:927-942), the re-composition E = x0 + W X0 on the device (hm_recompose).

    python tests/tools/ies_iterate_timing.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.update import IlesPlan, center, ies_step, recompose  # noqa: E402

N, n, nTime = 1000, 128, 40
_, gm = make_models(n, n)
prior = perms(n, n, N, seed=1)
X0, x0 = center(prior, dtype=64)
rng = np.random.RandomState(0)
W = np.eye(N) + 0.05 * rng.randn(N, N) / np.sqrt(N)
t0 = time.perf_counter()
E = recompose(W, X0, x0, dtype=64).astype(float)
t_rec = time.perf_counter() - t0
plan = ForwardPlan(gm, N, 0.025, nTime, keep_history=False)
plan.set_inputs(E, transformed=False)
plan.run()
plan.sync()
t0 = time.perf_counter()
plan.set_inputs(E, transformed=False)
plan.run()
st = plan.sync()
_, prods, status = plan.outputs(want_wsats=False)
t_fwd = time.perf_counter() - t0
plan.close()
Eo = prods.reshape(N, -1).astype(float)
innov = rng.randn(N, Eo.shape[1])
out = {}
for form in ("gram", "svd"):
    ies_step(W, Eo, innov, form)
    t0 = time.perf_counter()
    out[form] = ies_step(W, Eo, innov, form)
    print(f"host subspace algebra, {form:4s}: {time.perf_counter() - t0:.3f} s", flush=True)
print(f"max |gram - svd| = {np.abs(out['gram'] - out['svd']).max():.2e} (scale {np.abs(out['svd']).max():.2e})")
# the same step on the device (update.ies(subspace="device")): hm_iles_step with one domain and a taper of ones, from the identity weights
dplan = IlesPlan(prior, [np.arange(prior.shape[1])], np.ones((1, Eo.shape[1])), cutoff=0.5)
Sd = Eo - Eo.mean(0)
dplan.step(Sd, innov, 0.0)  # (a step of length 0: warm-up, the weights stay the identity)
t0 = time.perf_counter()
dplan.step(Sd, innov, 1.0)
Wd = dplan.weights(0)
t_dev = time.perf_counter() - t0
ref = np.eye(N) + ies_step(np.eye(N), Eo, innov, "gram")
print(f"device subspace algebra (hm_iles_step, one domain, incl. the copy of the {N} x {N} weights back): {t_dev:.3f} s; "
      f"max |device - host gram| = {np.abs(Wd - ref).max():.2e}")
dplan.close()
print(f"forward pass (host call, upload + 40 steps + producer series back): {t_fwd:.3f} s, device {st['ms_total'] / 1e3:.3f} s; "
      f"re-composition x0 + W X0 (host call): {t_rec:.3f} s; status ok: {not status.any()}")
