"""Device-to-device copy rate at the size of one attention launch (read 288 MB + write 288 MB at 5x4,
1000 frames): the practical ceiling of any kernel that reads ft2 once and writes the layer output once."""
import json
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 180000 * 400
x = torch.randn(n, device='cuda')
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 50
ev[0].record()
for _ in range(reps):
    y.copy_(x)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
print(json.dumps({'bytes_moved': 8 * n, 'us_per_copy': us, 'TB_per_s': 8 * n / us / 1e6}))
