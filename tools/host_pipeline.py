"""Host-inclusive rate with pinned staging and several engines on their own streams (dev tool; bench.py's
host_inclusive_pipelined is the same code).  Usage: python tools/host_pipeline.py [chunks] [passes] [warm passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for c in ([chunks] if len(sys.argv) > 1 else [1, 2, 4, 8, 16]):
    for co in (False, True):
        r = bench.host_inclusive_pipelined(np, torch, dev, chunks=c, passes=passes, warm=warm, controls_only=co)
        print(c, "chunks, %s: %.2f M solves/s, %.3f ms per batch" % ("[U | status | iter] out" if co else "whole result out",
                                                                     r["solves_per_s"] / 1e6, r["ms_per_batch"]))
