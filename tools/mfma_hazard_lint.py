#!/usr/bin/env python3
"""command line of copra_amd/hazard_lint.py (see there):   python tools/mfma_hazard_lint.py --library [libcopra_hip.so] | [file.s ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd.hazard_lint import *  # noqa: F401,F403,E402
from copra_amd.hazard_lint import CSRC, main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
