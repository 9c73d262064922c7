import torch
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
for fn, name, nb in ((lambda: y.copy_(x), "copy 1 GiB (R+W)", 2 << 30), (lambda: x.zero_(), "memset 1 GiB (W)", 1 << 30), (lambda: x.view(torch.float32).sum(), "sum 1 GiB fp32 (R)", 1 << 30)):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); print(f"{name}: {ts[3]*1e3:.1f} us -> {nb / ts[3] / 1e9:.2f} TB/s")
