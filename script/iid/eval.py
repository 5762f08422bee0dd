#!/usr/bin/env python
"""Metrics evaluation with the reference's benchmark protocol (script/iid/eval.py there: same flags, same files),
see marigold_amd/evaluation/harness.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marigold_amd.evaluation.harness import eval_main  # noqa: E402

if __name__ == "__main__":
    sys.exit(eval_main("iid"))
