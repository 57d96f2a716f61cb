#!/usr/bin/env python3
"""Dev: what the other FORMS of the extend operator cost on the config-3 chunk -- the one-stage kernel of deterministic
inference, and score_mod = relative_bias_score_mod (extent 1024) on both forms.  Prints bench.py's extend_forms_bench leg."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.extend_forms_bench(torch.device("cuda:0")), indent=1))
