"""The bench line's contract (task prompt, section 4): the committed line of the last GPU run (profiles/rNN/bench_default.json, written
by `python3 bench.py` on the MI355X box through profiles/tools/collect_rNN.sh) has every key the driver and the judge read, with the
types and relations they rely on.  CPU-only: it checks the record, not the GPU."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _latest_line():
    files = sorted(ROOT.glob("profiles/r*/bench_default.json"))
    assert files, "no committed bench line"
    return json.loads(files[-1].read_text()), files[-1]


def test_committed_bench_line_has_the_contract_keys():
    d, path = _latest_line()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                     ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), (path, key)
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    baseline = json.loads((ROOT / "BASELINE.json").read_text())
    assert d["metric"] == baseline["metric"] or d["metric"] in str(baseline)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "stale_inputs"):
        assert key in r, key
    # `frac` is always a number (this run's launch time priced with the committed instruction counts); staleness of those counts is a flag
    assert isinstance(r["frac"], float) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert isinstance(r["stale_inputs"], bool)
    if "frac_valid" in r:  # (lines from round 5 on) the committed line must be priced with counts taken from the objects that ran
        assert r["frac_valid"] is True and r["stale_inputs"] is False, (path, r.get("stale_inputs_detail"))
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # value = members x time steps x steps / wall, with the wall the line itself reports
    members, n_time = d["config"]["members_per_gpu"], d["config"]["nTime"]
    assert abs(d["value"] - members * n_time / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if "blocks" in d:  # (lines from round 5's second half on) the timed region runs the ensemble as member blocks on streams of their own
        b = d["blocks"]
        assert b["n"] >= 1 and b["bounds"][0] == 0 and b["bounds"][-1] == members and len(b["bounds"]) == b["n"] + 1
        if b["n"] > 1:
            one = b["one_block"]
            assert one["producer_series_identical_to_the_blocked_run"] is True and one["value"] > 0
            assert "one-block pass" in r["launch_times_measured_in"]
            # the blocked run is the faster one (that is why it is the default), but not by a margin that would point at skipped work
            assert one["value"] <= d["value"] <= 1.15 * one["value"]

    # (lines from round 6 on) silent fall-backs of the forward kernels are part of the line: the timed region's counters near the front,
    # and one record per leg that runs a forward model; a team retry inside the timed region makes the line's `ok` false
    if int(path.parent.name[1:]) >= 6:
        keys = {"nd_fallbacks", "team_retries", "slab_redos"}
        assert set(d["fallbacks_in_timed_region"]) == keys and d["fallbacks_in_timed_region"]["team_retries"] == 0
        assert set(d["blocks"]["fallbacks"]) == keys
        for leg in (d["config4"], d["config5"], d["es_update"]["es_mda_config3"]):
            assert set(leg["fallbacks"]) == keys and leg["fallbacks"]["team_retries"] == 0, leg
        sh = d["config2_strong_shard"]
        assert sh["members"] == 125 and sh["value"] > 0 and sh["status_ok"] is True and sh["fallbacks"]["team_retries"] == 0
        assert sh["value"] < d["value"] and abs(sh["implied_speedup_8_gpus"] - 8.0 * sh["value"] / d["value"]) < 1e-9  # beside `value`, never as it
        head = d["es_update_headline"]
        assert head["in_situ_ms"] == d["es_update"]["in_situ_ms"] and head["mfma_frac_in_situ"] == d["es_update"]["mfma_frac_in_situ"]
        assert list(d).index("es_update_headline") < list(d).index("roofline")  # in front of the long records: a truncated tail keeps it


def test_block_statistics_are_member_weighted():
    """forward.merge_block_stats: times are the longest block's, counts add up, means are weighted by the blocks' sizes (three blocks of
    333 / 333 / 334 members: the pairwise (a + b) / 2 fold of round 5 weighted the last block by a half)."""
    from historymatching_amd.forward import merge_block_stats

    sts = [dict(ms_total=3.0, mean_nts=600.0, member_steps=10, team_retries=0), dict(ms_total=5.0, mean_nts=630.0, member_steps=20, team_retries=1),
           dict(ms_total=4.0, mean_nts=615.0, member_steps=30, team_retries=0)]
    st = merge_block_stats(sts, [100, 100, 200])
    assert st["ms_total"] == 5.0 and st["member_steps"] == 60 and st["team_retries"] == 1 and st["blocks"] == 3
    assert abs(st["mean_nts"] - (600.0 * 100 + 630.0 * 100 + 615.0 * 200) / 400) < 1e-12


def test_member_blocks_are_multiples_of_32_members():
    """forward.block_bounds: the blocks a large ensemble is split into have a multiple of 32 members each (the last takes the rest), so that a
    member's workgroups meet the same XCD's L2 launch after launch; they cover the ensemble, in order, without gaps."""
    from historymatching_amd.forward import block_bounds

    assert block_bounds(1000, 3) == [0, 320, 640, 1000] and block_bounds(1000, 1) == [0, 1000] and block_bounds(512, 2) == [0, 256, 512]
    for N in (40, 100, 333, 768, 1000, 1024, 2000, 4096):
        for blocks in (1, 2, 3, 4):
            b = block_bounds(N, blocks)
            assert b[0] == 0 and b[-1] == N and len(b) == blocks + 1 and all(x < y for x, y in zip(b[:-1], b[1:])), (N, blocks, b)
            if blocks > 1 and N >= 64 * blocks:
                assert all((y - x) % 32 == 0 for x, y in zip(b[:-2], b[1:-1])), (N, blocks, b)
