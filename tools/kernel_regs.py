#!/usr/bin/env python3
"""VGPRs / AGPRs / scratch bytes per lane of every kernel in a built libwbcqp*.so, read from the code object's own metadata (no rebuild).
   python tools/kernel_regs.py [lib.so]        -- the build (inria_wbc_amd/build.py) refuses AGPRs and scratch in the solve kernels; this shows
   what a library on disk really holds, e.g. one built by tools/variants.sh, which does not pass through those checks."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co,
                        "--unbundle"], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in notes.split("- .agpr_count:")[1:]:
        rows.append((int(re.search(r"\.vgpr_count:\s+(\d+)", k).group(1)), int(k.split()[0]), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1)),
                     int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", k).group(1)), re.search(r"\.name:\s+(\S+)", k).group(1)))
    names = subprocess.run(["c++filt"], input="\n".join(r[4] for r in rows), capture_output=True, text=True).stdout.splitlines()
    return [(v, a, s, g, n) for (v, a, s, g, _), n in zip(rows, names)]


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "inria_wbc_amd", "lib", "libwbcqp.so")
    print("%5s %5s %8s %8s  kernel   (vgpr_count is the unified total: AGPRs included)" % ("VGPR", "AGPR", "scratch", "LDS"))
    for v, a, s, g, n in kernels(lib):
        print("%5d %5d %8d %8d  %s" % (v, a, s, g, n.split("(")[0]))
