#!/usr/bin/env python3
"""Dumps a synthetic batch for the C++ harnesses (inria_wbc::controllers::FileSource): 14 int64 header
(magic 0x5742435151, batch, 12 field lengths) followed by the 12 wbcqp_inputs fields as raw [B][len] doubles (the reader also
takes the older 13-word header with eleven fields, magic 0x5742435150: files written before the cop task's rows existed)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inria_wbc_amd import structure, synth  # noqa: E402


def dump(path, st, inputs):
    B = inputs["h"].shape[0]
    L = st.field_lengths()
    with open(path, "wb") as f:
        np.array([0x5742435151, B] + [L[k] for k in synth.FIELDS], dtype=np.int64).tofile(f)
        for k in synth.FIELDS:
            np.ascontiguousarray(inputs[k], dtype=np.float64).tofile(f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="talos")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seed-offset", type=int, default=0)
    ap.add_argument("out")
    a = ap.parse_args()
    st = structure.STRUCTURES[a.robot]()
    dump(a.out, st, synth.generate(st, a.batch, synth.SEED_BASE[a.robot] + a.seed_offset))
    print("wrote", a.out)
