#!/usr/bin/env python3
"""Static instruction census of a kernel's hot loop from the built object (no GPU needed).

    python3 profiles/tools/isa_count.py historymatching_amd/csrc/sat128.o k_sat128ILb1 32 2
    (object, substring of the mangled kernel name, cells per thread per loop trip, s_barrier count of the loop wanted)

Extracts the gfx950 code object from the fat object (llvm-objdump --offloading), disassembles it, finds the loop with the
largest body inside the first function whose mangled name contains the given substring (a backward s_cbranch to an earlier
address) and counts its instructions by class.  For k_sat128 that loop is the explicit sub-step loop: one trip = one sub-step of
the thread's 8 x 4 cell patch, so `dp_valu / 32` is the double-precision VALU instructions per cell per sub-step that
bench.py's roofline uses (profiles/rNN/isa_counts.json)."""
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        local = Path(d) / Path(obj).name
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local.name], cwd=d, check=True, capture_output=True)
        co = next(Path(d).glob("*gfx950*"))
        return subprocess.run([OBJDUMP, "-d", str(co)], check=True, capture_output=True, text=True).stdout


def functions(text):
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                out[name] = body
            name, body = m.group(1), []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and name:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if name:
        out[name] = body
    return out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_.*_f64", op) or op in ("v_rcp_f64_e32", "v_rcp_f64_e64"):
        return "dp_valu"
    if op.startswith("v_"):
        return "other_valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def hot_loop(body, barriers=None):
    """Largest backward-branch region [target, branch]; with `barriers` only regions holding exactly that many s_barrier
    (the compiler places cold blocks at the end of the function that jump BACK into the body: those regions span
    several loops and are told apart by their barrier count)."""
    addr_index = {a: i for i, (a, _, _) in enumerate(body)}
    best = best_key = None
    for i, (a, op, args) in enumerate(body):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", args) or re.search(r"<[^>+]*>", args)
        # objdump prints the target as  <func+0xOFF>; recover the absolute address from the simm16
        m16 = re.match(r"(\d+)", args.strip())
        if not m16:
            continue
        off = int(m16.group(1))
        if off >= 0x8000:
            off -= 0x10000
        target = a + 4 + 4 * off
        if target <= a and target in addr_index:
            j = addr_index[target]
            if barriers is not None and sum(1 for _, o, _ in body[j:i + 1] if o == "s_barrier") != barriers:
                continue
            # a region that reloads spilled registers is not the steady-state loop: where a wave wakes up from its dry phase the compiler
            # re-enters the loop through a block of scratch loads (and the prologue's flux scaling) that a back edge also spans
            scratch = any(o.startswith("scratch_") for _, o, _ in body[j:i + 1])
            cand = (not scratch, i - j)
            if best is None or cand > best_key:
                best, best_key = (j, i), cand
    return best


def main():
    obj, sub = sys.argv[1], sys.argv[2]
    per = float(sys.argv[3]) if len(sys.argv) > 3 else 32.0
    barriers = int(sys.argv[4]) if len(sys.argv) > 4 else None
    fns = functions(disassemble(obj))
    name = next(n for n in fns if sub in n)
    body = fns[name]
    lo, hi = hot_loop(body, barriers)
    # blocks the compiler laid out behind the loop's back edge that are entered from the loop and jump back into it (k_sat128r: the
    # fractional flow of the two published rows, the not-dry side of a wave-uniform branch) belong to the trip
    def target_of(a, args):
        m16 = re.match(r"(\d+)", args.strip())
        if not m16:
            return None
        off = int(m16.group(1))
        return a + 4 + 4 * (off - 0x10000 if off >= 0x8000 else off)
    a_lo, a_hi = body[lo][0], body[hi][0]
    tail = []
    e = hi + 1
    while e < len(body):
        a, op, args = body[e]
        if op == "s_branch":
            t = target_of(a, args)
            if t is not None and a_lo <= t <= a_hi:
                entered = any(o.startswith(("s_cbranch", "s_branch")) and (target_of(x, g) or 0) > a_hi and (target_of(x, g) or 0) <= a
                              for x, o, g in body[lo:hi + 1])
                if entered:
                    tail = list(range(hi + 1, e + 1))
            break
        if op in ("s_endpgm",):
            break
        e += 1
    counts = {}
    rare = {}
    divisions = 0
    # short blocks skipped by a forward scalar branch (k_sat128r: the injector's addend, taken on one patch row of one wave) are
    # counted apart: a trip that does not take them executes none of their instructions
    skip_until = -1
    addr = [a for a, _, _ in body]
    for idx in list(range(lo, hi + 1)) + tail:
        a, op, args = body[idx]
        if idx <= skip_until:
            rare[classify(op)] = rare.get(classify(op), 0) + 1
            continue
        c = classify(op)
        counts[c] = counts.get(c, 0) + 1
        if op.startswith("v_rcp_f64"):  # one reciprocal per division (compiler's IEEE sequence or fracflow.h's unscaled form)
            divisions += 1
        if op in ("s_cbranch_scc0", "s_cbranch_scc1"):
            m16 = re.match(r"(\d+)", args.strip())
            off = int(m16.group(1)) if m16 else 0
            if 0 < off <= 24:
                target = a + 4 + 4 * off
                skip_until = max(i for i in range(idx, hi + 1) if addr[i] < target)
    out = {"object": obj, "kernel": name, "loop_instructions": hi - lo + 1 + len(tail), "out_of_line_instructions": len(tail), "loop_start": hex(body[lo][0]), "loop_end": hex(body[hi][0]), "counts": counts, "rarely_taken_blocks": rare, "fp64_divisions": divisions,
           "cells_per_thread_per_trip": per, "dp_valu_per_cell_substep": counts.get("dp_valu", 0) / per,
           "valu_per_cell_substep": (counts.get("dp_valu", 0) + counts.get("other_valu", 0)) / per}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
