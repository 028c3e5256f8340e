#!/usr/bin/env python3
"""Regenerates the frozen vectors in this directory from the CPU oracle.

The reference itself cannot be imported in the build container (its dependencies are
absent), so these are NOT reference outputs: they freeze the oracle's results on small
seeded inputs, so that a later change to the oracle or to the HIP library that moves a
result is noticed.  Inputs are regenerated from the seeds; only expected outputs are
stored.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from golden_cases import CASES  # noqa: E402


def main():
    for name, fn in CASES.items():
        out = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {sum(v.nbytes for v in out.values())} B -> {os.path.getsize(path)} B on disk")


if __name__ == "__main__":
    main()
