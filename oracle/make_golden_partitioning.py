"""Generate tests/golden/f7_partitioning.npz from the REAL reference helper `rectangular_partitioning`
(notebooks/tools/localization.py:95-145), imported from /root/reference (build container only; never copied).
Ragged batch lists are stored as one concatenated index vector plus batch lengths.

Usage:  python oracle/make_golden_partitioning.py
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import OUT, import_reference  # noqa: E402

CASES = [((4, 13), (2, 4)), ((20, 20), (5, 5)), ((7,), (3,)), ((6, 5, 4), (2, 2, 3)), ((128, 128), (16, 32))]


def main():
    _, loc, _, _ = import_reference()
    out = {}
    for c, (shape, steps) in enumerate(CASES):
        batches = loc.rectangular_partitioning(list(shape), list(steps))
        out[f"c{c}_shape"] = np.array(shape)
        out[f"c{c}_steps"] = np.array(steps)
        out[f"c{c}_len"] = np.array([len(b) for b in batches])
        out[f"c{c}_ind"] = np.concatenate(batches)
        sub = loc.rectangular_partitioning(list(shape), list(steps), do_ind=False)
        out[f"c{c}_sub"] = np.concatenate([np.stack(b, 0) for b in sub], axis=1)
    np.savez_compressed(OUT / "f7_partitioning.npz", **out)
    print("f7_partitioning.npz", (OUT / "f7_partitioning.npz").stat().st_size)


if __name__ == "__main__":
    main()
