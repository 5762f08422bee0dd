#!/usr/bin/env python
"""Depth inference over a folder of images - same flags and output files as the reference's
script/depth/run.py, running on the MI355X engine (see marigold_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marigold_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main("depth"))
