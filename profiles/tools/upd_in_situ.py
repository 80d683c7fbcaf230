"""Per-kernel durations of the analysis step in situ (inside the 4-pass ES-MDA of bench.py: each step right behind a forward pass) against
the same step queued back to back, from a rocprofv3 kernel trace of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config4 --no-config5
--no-two-streams --no-host-call`:   python3 profiles/tools/upd_in_situ.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ("k_center_gram", "k_ldl_chain", "k_gxt_dma", "k_apply_dma", "k_cast")
short = lambda n: next((s for s in names if s in n), None)  # noqa: E731
steps, cur = [], None
for i, r in enumerate(rows):
    s = short(r["Kernel_Name"])
    if s is None:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if s == "k_center_gram":
        # right behind a forward pass: a sweep or pressure kernel ended within 200 us before this step's first kernel (runtime fill / copy /
        # cast kernels may sit in between)
        t0 = int(r["Start_Timestamp"])
        near = [q for q in rows[max(0, i - 8):i] if t0 - int(q["End_Timestamp"]) < 200000 and ("k_sat" in q["Kernel_Name"] or "k_nd" in q["Kernel_Name"])]
        cur = {"after": "forward pass" if near else "update", "t0": t0}
        steps.append(cur)
    if cur is not None and s != "k_cast":
        cur[s] = dur
        cur["t1"] = int(r["End_Timestamp"])
for kind in ("forward pass", "update"):
    sel = [s for s in steps if s["after"] == kind and all(k in s for k in names[:4]) and s["k_gxt_dma"] < 100.0]  # (config 3's shape)
    if not sel:
        continue
    sel = sel[-40:] if kind == "update" else sel
    print(f"analysis steps right behind a{'n' if kind == 'update' else ''} {kind}: {len(sel)}")
    for k in names[:4]:
        v = sorted(s[k] for s in sel)
        print(f"  {k:14s} median {v[len(v) // 2]:7.1f} us   min {v[0]:7.1f}   max {v[-1]:7.1f}")
    v = sorted((s["t1"] - s["t0"]) / 1e3 for s in sel)
    print(f"  first start to last end: median {v[len(v) // 2]:7.1f} us")
