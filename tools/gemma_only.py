#!/usr/bin/env python3
"""GPU box: BASELINE config 4's probes ALONE (no headline engine): `text` = Gemma-3-4B text tower, `image` = image + text.  Run under
rocprofv3 --kernel-trace (tools/gemma_profile.sh) for clean per-kernel figures.   Usage: gemma_only.py [text|image] [steps] [nosite]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
import lxt_amd.ops as ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "text"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.SITE_FUSION = "nosite" not in sys.argv
dev = torch.device("cuda:0")
text, mm = bench.config4_probe(ops, dev, torch.bfloat16, 2500.0, steps=steps, image=(what == "image"))
print(json.dumps({"site_fusion": ops.SITE_FUSION, "text": {k: v for k, v in text.items() if k != "workload"},
                  "image_text": None if mm is None else {k: v for k, v in mm.items() if k != "workload"}}))
