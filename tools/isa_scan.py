#!/usr/bin/env python3
"""ISA-level checks of the HIP kernels (no GPU needed; hipcc cross-compiles gfx950):

    python tools/isa_scan.py k_decgemm.hip [kernel-name-substring]

For every kernel of the file: total instructions, the instruction index of the first global load, VGPRs, scratch bytes, and the number
of "serialised loads" — a global/buffer load followed within three instructions by `s_waitcnt vmcnt(0)`, which is what a load
under a wave-uniform branch compiles to (issued and awaited alone).  NOTES.md, round 2, explains the two compiler behaviours
this looks for."""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bioreason_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result".split()


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    if not os.path.exists(src):
        src = os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-S", "--cuda-device-only", src, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    kern, rows, meta = None, {}, {}
    recent = []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, recent = m.group(1), []
            rows[kern] = dict(n=0, first=0, serial=0)
            continue
        m = re.match(r"^\s*\.amdhsa_kernel (\S+)", l)
        if m:
            cur = m.group(1)
            meta[cur] = {}
        m = re.match(r"^\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size) (\d+)", l)
        if m and meta:
            meta[cur][m.group(1)] = int(m.group(2))
        if kern is None:
            continue
        if l.startswith(".Lfunc_end"):
            kern = None
            continue
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        r = rows[kern]
        r["n"] += 1
        if not r["first"] and (s.startswith("global_load") or s.startswith("buffer_load")):
            r["first"] = r["n"]
        if s.startswith("s_waitcnt") and "vmcnt(0)" in s and any(p.startswith(("global_load", "buffer_load")) for p in recent[-3:]):
            r["serial"] += 1
        recent.append(s)
    names = demangle(list(rows))
    print("%6s %6s %6s %5s %7s  kernel" % ("instr", "1st-ld", "serial", "vgpr", "scratch"))
    for k, r in rows.items():
        d = names.get(k, k)
        if pat and pat not in d:
            continue
        mt = meta.get(k, {})
        print("%6d %6d %6d %5s %7s  %s" % (r["n"], r["first"], r["serial"], mt.get("next_free_vgpr", "?"),
                                           mt.get("private_segment_fixed_size", "?"), d[:150]))


if __name__ == "__main__":
    main()
