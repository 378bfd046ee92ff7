#!/usr/bin/env python3
"""Writes tests/golden/oracle_checksums.json from the CPU oracle (seeded inputs; see
tests/test_oracle_cpu.py::compute_checksums). Re-run only when the oracle's defined semantics change."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle  # noqa: E402
from test_oracle_cpu import compute_checksums  # noqa: E402

json.dump(compute_checksums(oracle), open(os.path.join(HERE, "oracle_checksums.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(HERE, "oracle_checksums.json")).read())
