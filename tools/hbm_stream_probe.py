import torch, time
dev = torch.device("cuda:0")
x = torch.empty((1, 32, 64, 384, 768), dtype=torch.float32, device=dev)
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
gb = x.numel() * 4 / 1e9
a = t(lambda: x.fill_(1.0)); print("fill  %.3f ms  %.0f GB/s write" % (a * 1e3, gb / a))
a = t(lambda: x.zero_()); print("zero  %.3f ms  %.0f GB/s write" % (a * 1e3, gb / a))
a = t(lambda: y.copy_(x)); print("copy  %.3f ms  %.0f GB/s (read+write %.0f)" % (a * 1e3, gb / a, 2 * gb / a))
a = t(lambda: torch.mul(x, 2.0, out=y)); print("mul   %.3f ms  %.0f GB/s (read+write %.0f)" % (a * 1e3, gb / a, 2 * gb / a))
