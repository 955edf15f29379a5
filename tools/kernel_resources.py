#!/usr/bin/env python3
"""Registers, LDS and occupancy clang reports for every kernel of libwbcqp.so (compile only; no GPU needed).
Usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inria_wbc_amd import build  # noqa: E402


def main():
    cmd = [build.hipcc(), "-Rpass-analysis=kernel-resource-usage", "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={build.ARCH}",
           "-fno-gpu-rdc", "-ffp-contract=on", *sys.argv[1:], os.path.join(build.CSRC, "wbcqp_api.hip"), "-o", "/tmp/_res.so", "-ldl"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    for name, u in build._resource_usage(out).items():
        short = name[:100]
        print("%-100s VGPR %3d AGPR %3d SGPR %3d scratch %3d occ %d" % (short, u.get("VGPRs", -1), u.get("AGPRs", -1), u.get("SGPRs", -1),
              u.get("ScratchSize [bytes/lane]", -1), u.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
