"""The round-3 finding on THIS code base: TrainEngine(use_graph=True) captures its step, the engine dies and (MSAU_GRAPH_DESTROY=1) its
graphs are destroyed, the next engine captures again -- crashed in about half of the stand-alone runs of tests/test_train_gpu.py.
    MSAU_GRAPH_DESTROY=1 [MSAU_CAPTURE_FRESH_EVENTS=1] python tools/repro/graph_destroy_msau.py [cycles]"""
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
print(f"PASS {cycles} engine cycles (graphs destroyed: {os.environ.get('MSAU_GRAPH_DESTROY', '0')}, fresh events: {os.environ.get('MSAU_CAPTURE_FRESH_EVENTS', '0')})")
