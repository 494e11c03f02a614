"""GPU box: device-to-device copy bandwidth (read + write bytes / time), the practical HBM ceiling next to the 8 TB/s vendor figure"""
import json, sys, torch
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
a.fill_(1.0)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"d2d_copy_GBps_read_plus_write": 2 * 4 * n / (ms * 1e-3) / 1e9, "bytes": 4 * n, "ms": ms}))
