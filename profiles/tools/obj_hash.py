#!/usr/bin/env python3
"""sha256 of the built objects a profile refers to -- recorded in isa_counts.json / fp64_roofline.json when a profile is
taken, compared by bench.py with the objects it runs (roofline.stale_inputs):

    python3 profiles/tools/obj_hash.py            -> {"sat128.o": "...", "press128s.o": "...", ...}"""
import hashlib
import json
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parents[2] / "historymatching_amd" / "csrc"
OBJECTS = ("sat128r.o", "sat128.o", "press128s.o", "press_nd.o")


def object_hashes(names=OBJECTS):
    out = {}
    for n in names:
        f = CSRC / n
        out[n] = hashlib.sha256(f.read_bytes()).hexdigest() if f.exists() else None
    return out


if __name__ == "__main__":
    json.dump(object_hashes(sys.argv[1:] or OBJECTS), sys.stdout, indent=1)
    print()
