#!/usr/bin/env python
"""Intrinsic-image-decomposition inference over a folder of images - same flags and output files as the reference's
script/iid/run.py, running on the MI355X engine (see marigold_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marigold_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main("iid"))
