"""Host -> device rate for uint16 frames (pinned memory, one stream): what a caller that hands over host buffers pays."""
import time
import torch
n = 2048 * 270
host = torch.empty((n, 120, 160), dtype=torch.int16).pin_memory()
dev = torch.empty_like(host, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("pinned H2D: %.1f GB in %.3f s = %.1f GB/s = %.2f M frames/s" % (host.numel() * 2 / 1e9, dt, host.numel() * 2 / 1e9 / dt, n / dt / 1e6))
