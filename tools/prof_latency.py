"""Batch-1 tower + NCI generate, eager, a few repetitions (for rocprofv3 --kernel-trace --stats): which kernels make up the
single-query latency."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth
dev = torch.device("cuda:0")
model, tower, _, _ = synth.build(dev, 4, 32, None)
tower.batch_size = None
ids, mask = synth.query_ids(64, dev, np.random.default_rng(0))
what = sys.argv[1] if len(sys.argv) > 1 else "tower"
for i in range(10):
    if what == "tower":
        tower.encode_query({"input_ids": ids[i:i + 1], "attention_mask": mask[i:i + 1]}, graph=False, pack=False) if False else \
            tower.encode_query({"input_ids": ids[i:i + 1], "attention_mask": mask[i:i + 1]})
    else:
        model.generate(ids[i:i + 1], mask[i:i + 1], num_beams=10)
torch.cuda.synchronize()
