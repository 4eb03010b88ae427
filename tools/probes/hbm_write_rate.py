import torch, sys
sys.path.insert(0, "/root/repo")
import bench
x = torch.empty(414_515_200 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill (write only)", lambda: x.fill_(1.5)), ("copy (read + write)", lambda: y.copy_(x)), ("sum (read only)", lambda: x.sum())):
    us = bench.event_time_us(fn, 20)
    nbytes = x.numel() * 4 * (2 if "copy" in name else 1)
    print(f"{name}: {us:.1f} us for {nbytes/1e6:.0f} MB = {nbytes/us/1e6:.2f} TB/s")
