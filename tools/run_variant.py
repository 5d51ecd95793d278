"""Runs one of bench.py's secondary workloads on its own (profiling aid): python tools/run_variant.py scripts|distill|model40k|sa [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "scripts"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
r = bench.run_extra(kind, torch.device("cuda:0"), steps, 3)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "steps")}))
