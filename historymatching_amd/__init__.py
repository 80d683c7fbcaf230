"""historymatching_amd -- MI355X-native ensemble forward model + ensemble-smoother update.

Drop-in for the data-parallel hot path of patnr/HistoryMatching:
``ressim.ResSim`` (simulator object), ``forward.make_forward_model`` (= the notebook's ``forward_model``),
``update.ens_update0`` / ``update.ens_update0_loc`` / ``update.center``.  All compute goes through the
C ABI of ``libhm_amd.so`` (hand-written HIP for gfx950); there is no CPU fallback.
"""

__all__ = ["ressim", "forward", "update", "localization", "geostat", "dist"]
