"""The round-3 finding on THIS code base: TrainEngine(use_graph=True) captures its step, the engine dies, its graphs are destroyed,
the next engine captures again -- crashed in about half of the stand-alone runs of tests/test_train_gpu.py while the captured
backward forked its weight gradients onto the side stream (profiles/r04_graph_destroy.md: 4 of 8 runs; 0 of 8 on one stream).
Since round 4 every captured sweep is single-stream (Plan.forward / Plan.backward detect the capture themselves), so as shipped
this script PASSES; it is the regression check of that remedy.  The crashing configuration cannot be reached through the
product any more -- to see it again, remove the `is_current_stream_capturing()` line from Plan.backward.
    python tools/repro/graph_destroy_msau.py [cycles]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from msau_amd.model import MSAUWrapper, TrainEngine

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype="bf16", seed=0)
torch.manual_seed(0)
x = torch.randn(2, 16, 64, 48, device="cuda")
lab = torch.randint(0, 5, (2, 64, 48), device="cuda")
m = MSAUWrapper(16, 5, kw).cuda()
for c in range(cycles):
    eng = TrainEngine(m, use_graph=True)
    for _ in range(3):
        loss = eng.step(x, lab)
    torch.cuda.synchronize()
    eager = TrainEngine(m)                      # eager sweeps between the captures use the same event pool
    eager.step(x, lab)
    torch.cuda.synchronize()
    del eng, eager
    gc.collect()
    print(f"cycle {c + 1} ok loss {float(loss):.4f}", flush=True)
print(f"PASS {cycles} engine cycles (single-stream captures, graphs destroyed with their engines)")
