"""List, in order, the kernels one configs[3]-shaped PaletteNeRF training step launches (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import ProfilerActivity, profile
kind = sys.argv[1] if len(sys.argv) > 1 else "palette"
dev = torch.device("cuda:0")
m, step = bench.make_training_step(kind, 4096, dev)
for i in range(6):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    step(7)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
evs.sort(key=lambda e: e.time_range.start)
print(len(evs), "kernels")
for e in evs:
    print(f"{e.device_time:8.1f}  {e.name[:110]}")
