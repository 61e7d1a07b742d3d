"""How fast is one pinned H2D copy of a JSON window (55 MB) on this box: alone, and while fp32 GEMMs keep the CUs busy on
another stream?  Separates "PCIe is the limit" from "the copy waits for compute".  python tools/h2d_probe.py [MB]"""
import sys, time
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 55
h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(mb << 20, dtype=torch.uint8, device='cuda')
a = torch.randn(8192, 8192, device='cuda'); b = torch.randn(8192, 8192, device='cuda')
s_copy = torch.cuda.Stream(priority=-1); s_cmp = torch.cuda.Stream()
def copy_ms(busy):
    out = []
    for _ in range(6):
        if busy:
            with torch.cuda.stream(s_cmp):
                for _ in range(4): a @ b
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_copy):
            e0.record(); d.copy_(h, non_blocking=True); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return min(out[1:]), max(out[1:])
for busy in (False, True, False):
    lo, hi = copy_ms(busy)
    print('%d MB pinned H2D, compute %s: %.2f .. %.2f ms = %.1f GB/s' % (mb, 'busy' if busy else 'idle', lo, hi, (mb << 20) / lo / 1e6))
