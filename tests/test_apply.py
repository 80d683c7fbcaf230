"""``utils.apply`` of the host mirror against the reference's map semantics (tools/utils.py:155-242): member-wise zip of positional
and keyword ensembles, member order, strict lengths, and -- on the GPU -- the batched forms of the hot-path functions."""
import numpy as np
import pytest

from tests.helpers import perms

DT, NT = 0.025, 8


def test_apply_maps_member_wise_in_order_with_keyword_ensembles_and_strict_lengths():
    from historymatching_amd import utils

    calls = []

    def fun(a, b, scale=1.0):
        calls.append((a, b))
        return (a + b) * scale

    A, B, S = np.arange(5.0), 10 * np.arange(5.0), np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    assert utils.apply(fun, A, B) == [11 * k for k in range(5)]
    assert calls == list(zip(A, B))                                    # one call per member, in member order (utils.py:175, 228-231)
    assert utils.apply(fun, A, B, scale=S, pbar=False) == [11 * k * (k + 1) for k in range(5)]   # keyword ensembles are zipped too (utils.py:172-173)
    rows = utils.apply(lambda row: row.sum(), np.ones((3, 4)))          # the 0th axis is the ensemble axis
    assert rows == [4.0, 4.0, 4.0]
    with pytest.raises(ValueError):                                    # zip(strict=True), utils.py:175
        utils.apply(fun, A, B[:4])
    with pytest.raises(ValueError):
        utils.apply(fun, A, B, scale=S[:2])
    utils.nCPU = "auto"                                               # the notebooks assign it (HistoryMatch.py:376-380); no effect here
    assert utils.apply(fun, A[:2], B[:2]) == [0.0, 11.0]
    utils.nCPU = 1


def test_apply_hands_the_whole_ensemble_to_a_batched_form_and_counts_the_calls():
    from historymatching_amd import utils

    seen = []

    def obj(u, w=0.0):
        raise AssertionError("the member-wise form must not run when a batched one exists")

    def batched(U, w=None):
        seen.append((len(U), None if w is None else len(w)))
        return [float(np.sum(u)) + (0.0 if w is None else wk) for u, wk in zip(U, np.zeros(len(U)) if w is None else w)]

    obj.batched = batched
    obj.nCalls = 0
    U = np.arange(12.0).reshape(4, 3)
    assert utils.apply(obj, U) == [3.0, 12.0, 21.0, 30.0]
    assert utils.apply(obj, U, w=np.array([1.0, 1.0, 1.0, 1.0])) == [4.0, 13.0, 22.0, 31.0]
    assert seen == [(4, None), (4, 4)] and obj.nCalls == 8              # utils.py:222-224
    assert utils.apply(obj, U[:0]) == []
    with pytest.raises(ValueError):
        utils.apply(obj, U, w=np.ones(3))
    obj.batched = lambda U: [0.0]
    with pytest.raises(RuntimeError):
        utils.apply(obj, U)


@pytest.mark.gpu
def test_apply_comp1_is_the_notebooks_forward_model_in_one_device_call():
    """HistoryMatch.py:383-387: ``forward_model`` is ``apply(comp1, *ens_args)`` transposed.  With the mirror's ``comp1`` the map
    is one device run; member k of the result equals ``comp1(perm_k)`` on its own, with and without start states."""
    from historymatching_amd import utils
    from historymatching_amd.forward import make_forward_model
    from tests.test_forward_gpu import make_models

    n, N = 20, 5
    _, gm = make_models(n, n)
    fm = make_forward_model(gm, DT, NT)
    comp1 = fm.comp1
    x = perms(n, n, N, seed=11)
    w0 = np.random.RandomState(2).uniform(0, 0.3, (N, n * n))
    for ens_args, kw in (((x,), {}), ((x, w0), {}), ((x,), {"wsat0": w0})):
        pairs = utils.apply(comp1, *ens_args, pbar=False, **kw)
        assert len(pairs) == N
        wsats, prods = (np.array(v) for v in zip(*pairs))                # the notebook's transpose (HistoryMatch.py:386-387)
        assert wsats.shape == (N, NT + 1, n * n) and prods.shape == (N, NT, 4)
        w_ref, p_ref = fm(*ens_args, **kw)
        assert np.array_equal(wsats, w_ref) and np.array_equal(prods, p_ref)
        for k in (0, N - 1):
            wk, pk = comp1(x[k]) if len(ens_args) + len(kw) == 1 else comp1(x[k], w0[k])
            assert np.array_equal(wk, wsats[k]) and np.array_equal(pk, prods[k])
    with pytest.raises(ValueError):
        utils.apply(comp1, x, w0[:3])


@pytest.mark.gpu
def test_apply_objective_values_an_ensemble_of_controls_in_one_run():
    """Optimise.py:441, 514: ``apply(obj, U)`` over an ensemble of injector positions; ``NpvBatch.as_objective`` gives ``obj`` its
    batched form.  Same values as the objective member by member."""
    from historymatching_amd import utils
    from historymatching_amd.opt import NpvBatch
    from tests.test_opt_gpu import _models

    _, gm = _models(20)
    batch = NpvBatch(gm, 0.025, 40)
    obj = batch.as_objective(lambda xy: dict(inj_xy=[list(xy)]))
    U = np.array([[0.3, 0.7], [1.0, 0.5], [1.9, 0.05], [2.5, 0.5]])      # the last one lies outside the domain -> 0 (Optimise.py:119-124)
    values = utils.apply(obj, U)
    assert len(values) == 4 and values[3] == 0 and all(abs(v) > 1 for v in values[:3])
    np.testing.assert_allclose(values, [obj(u) for u in U], rtol=1e-12, atol=1e-12)
