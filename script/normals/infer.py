#!/usr/bin/env python
"""Dataset inference with the reference's benchmark protocol (script/normals/infer.py there: same flags, same files),
see marigold_amd/evaluation/harness.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marigold_amd.evaluation.harness import infer_main  # noqa: E402

if __name__ == "__main__":
    sys.exit(infer_main("normals"))
