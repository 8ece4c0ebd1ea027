"""Where the HOST time of a small-batch pass goes (cProfile; the GPU is idle most of such a pass):
python tools/prof_host.py [queries] [tower|nci]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
what = sys.argv[2] if len(sys.argv) > 2 else "tower"
dev = torch.device("cuda:0")
model, tower, _, _ = synth.build(dev, 4, 32, None)
ids, mask = synth.query_ids(n, dev, np.random.default_rng(0))
q = {"input_ids": ids, "attention_mask": mask}
fn = (lambda: tower.encode_query(q)) if what == "tower" else (lambda: model.generate(ids, mask, num_beams=10))
for _ in range(3):
    fn()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    fn()
torch.cuda.synchronize()
print("%s, %d queries: %.3f ms per pass" % (what, n, (time.perf_counter() - t) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fn()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
