import torch, time
dev = torch.device('cuda', 0)
def bench(fn, nbytes, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return nbytes / ms / 1e6
for mib in (256, 1024, 4096):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.ones(n, dtype=torch.float32, device=dev); c = torch.ones(n, dtype=torch.float32, device=dev)
    print('%5d MiB per array: fill %.0f GB/s  copy %.0f GB/s (r+w)  triad %.0f GB/s (2r+w)' % (
        mib, bench(lambda: a.fill_(1.0), n * 4), bench(lambda: a.copy_(b), 2 * n * 4), bench(lambda: torch.add(b, c, alpha=3.0, out=a), 3 * n * 4)))
