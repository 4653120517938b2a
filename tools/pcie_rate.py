"""Raw PCIe copy rates (pinned / pageable, both directions) with torch, idle GPU.        python tools/pcie_rate.py"""
import torch, time
for mb in (64, 256, 1024):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    hp = torch.empty(n, dtype=torch.uint8).pin_memory()
    hu = torch.empty(n, dtype=torch.uint8)
    for name, src, dst in (("D2H pinned", d, hp), ("H2D pinned", hp, d), ("D2H pageable", d, hu)):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        print(f"{name:14s} {mb:5d} MiB: {3 * n / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
