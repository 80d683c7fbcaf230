"""Host-side mirror of ``tools/utils.py:apply`` -- the reference's ensemble map (utils.py:155-242).

The reference maps a per-member function over the 0th axis of its ensemble arguments, one process per core
(``utils.nCPU``).  Here the functions that sit on the hot path carry a *batched* form -- ``comp1`` of
``forward.make_forward_model`` (HistoryMatch.py:358-364, mapped at :383-387) and the objective of ``opt.NpvBatch``
(Optimise.py:112-125, mapped at :259, 441, 514, 655, 908) -- and ``apply`` hands them the whole ensemble in one device call;
what comes back is the list of per-member results the reference's callers transpose.  Any other function is host logic and
is called member by member, in order.  ``nCPU`` is kept as a module global because the notebooks set it; it has no effect.
"""
import numpy as np

nCPU = 1
"Kept for notebooks that assign it (utils.py:151); the ensemble is one device call whatever its value."


def apply(fun, *args, pbar=True, **kwargs):
    """``apply(fun, *ensembles, pbar=..., **named_ensembles)`` -> ``[fun(*members, **named_members), ...]`` in member order.

    Like the reference (utils.py:171-175) keyword ensembles are zipped member-wise together with the positional ones and
    ensembles of different lengths raise ``ValueError`` (``zip(strict=True)``).  ``pbar`` drove the reference's progress bar
    and is accepted and ignored.  ``fun.nCalls``, if present, is advanced by the number of members when the batched form ran
    (utils.py:222-224 does so for its process pool; the member-wise loop leaves the counting to ``fun`` itself)."""
    ensembles = list(args) + list(kwargs.values())
    if not ensembles:
        raise TypeError("apply() needs at least one ensemble argument")
    lengths = [len(e) for e in ensembles]
    if len(set(lengths)) > 1:
        k = next(i for i, n in enumerate(lengths) if n != lengths[0])
        raise ValueError(f"zip() argument {k + 1} is {'shorter' if lengths[k] < lengths[0] else 'longer'} than argument 1")
    batched = getattr(fun, "batched", None)
    if batched is not None:
        if lengths[0] == 0:
            return []
        named = dict(zip(kwargs, ensembles[len(args):]))
        out = list(batched(*args, **named))
        if len(out) != lengths[0]:
            raise RuntimeError(f"{getattr(fun, '__name__', fun)}.batched returned {len(out)} results for {lengths[0]} members")
        if hasattr(fun, "nCalls"):
            fun.nCalls += lengths[0]
        return out
    nPositional = len(args)
    out = []
    for members in zip(*ensembles):
        out.append(fun(*members[:nPositional], **dict(zip(kwargs, members[nPositional:]))))
    return out


def center(E, axis=0, rescale=False, **device_options):
    """``center(E, axis=0, rescale=False)`` -> ``(X, x)`` (utils.py:10-28) on the GPU: see ``update.center``."""
    from .update import center as _center

    return _center(E, axis=axis, rescale=rescale, **device_options)
