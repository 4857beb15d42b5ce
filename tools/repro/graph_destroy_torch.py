"""The same question through torch.cuda.CUDAGraph (what TrainEngine uses), still without any msau code: capture a few hundred torch
kernels on two streams, replay, DROP the graph (its destructor calls hipGraphExecDestroy / hipGraphDestroy), capture the next.
    python tools/repro/graph_destroy_torch.py [cycles]"""
import gc
import sys

import torch

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
y = torch.zeros_like(x)
z = torch.zeros_like(x)
gs = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
for c in range(cycles):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        y.add_(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=gs):
            for k in range(300):
                y.add_(x)
                if k % 6 == 5:
                    side.wait_stream(gs)
                    with torch.cuda.stream(side):
                        z.add_(x, alpha=2.0)
            gs.wait_stream(side)
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    del g
    gc.collect()
    if c % 10 == 9:
        print(f"cycle {c + 1} ok", flush=True)
print(f"PASS {cycles} capture / destroy cycles")
