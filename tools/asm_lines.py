#!/usr/bin/env python3
"""Static instruction counts of one kernel per source line, from an assembly file built with -gline-tables-only.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=on --cuda-device-only -gline-tables-only -S \
          -o k.s inria_wbc_amd/csrc/wbcqp_api.hip
    python tools/asm_lines.py k.s _ZN5wbcqp18solve_queue_kernelIdLb1ELi1E wbcqp_compact.hpp 939 1362

Prints, for the given file and line range, the instructions whose innermost .loc names that line, split into VALU / SALU / LDS /
VMEM / other (inlined callees are attributed to the line of the call through the `inlined_at` chain clang prints as comments is
NOT available in -S output, so a callee's instructions appear under the callee's own file:line -- the summary lists those files too).
"""
import re
import sys
from collections import defaultdict


def classify(op):
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    want_file = sys.argv[3] if len(sys.argv) > 3 else None
    lo = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    hi = int(sys.argv[5]) if len(sys.argv) > 5 else 10 ** 9
    files = {}
    counts = defaultdict(lambda: defaultdict(int))
    infn = False
    cur = (None, 0)
    with open(path) as f:
        for ln in f:
            s = ln.strip()
            m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
            if m:
                files[int(m.group(1))] = m.group(2)
                continue
            if not infn:
                if s.startswith(kern) and s.split(":")[0].startswith(kern) and ":" in s:
                    infn = True
                continue
            if s.startswith(".Lfunc_end"):
                break
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
            if m:
                cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
                continue
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                continue
            op = s.split()[0]
            counts[cur][classify(op)] += 1
    tot = defaultdict(int)
    perfile = defaultdict(lambda: defaultdict(int))
    for (fn, line), c in counts.items():
        for k, v in c.items():
            perfile[fn][k] += v
    print("per file:")
    for fn, c in sorted(perfile.items(), key=lambda kv: -sum(kv[1].values())):
        print("  %-28s %s" % (fn, dict(c)))
    if want_file:
        print("lines of %s in [%d, %d]:" % (want_file, lo, hi))
        for (fn, line), c in sorted(counts.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
            if fn == want_file and lo <= line <= hi:
                for k, v in c.items():
                    tot[k] += v
                print("  %5d  valu %4d salu %4d lds %3d vmem %3d" % (line, c["valu"], c["salu"], c["lds"], c["vmem"]))
        print("  total", dict(tot))


if __name__ == "__main__":
    main()
