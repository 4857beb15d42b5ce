import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, ctypes
from msau_amd import _lib
lib=_lib.load()
x=torch.zeros(1024,device='cuda')
s=torch.cuda.current_stream().cuda_stream
def run(n):
    for _ in range(n): _lib.call('msau_fill_zero', s, x.data_ptr(), 4096)
run(100); torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
# park the stream so host enqueue is hidden
_lib.call('msau_spin', s, 30000)
e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
print('back-to-back tiny kernel: %.2f us/launch (gpu timeline)' % (e0.elapsed_time(e1)*1000/2000))
t=time.perf_counter(); run(2000); t1=time.perf_counter(); torch.cuda.synchronize()
print('host enqueue: %.2f us/launch' % ((t1-t)*1e6/2000))
