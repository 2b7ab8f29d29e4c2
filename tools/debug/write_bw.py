import torch, time
x = torch.empty(914358272, dtype=torch.float64, device="cuda")
for name, fn in (("zero_", lambda: x.zero_()), ("fill_", lambda: x.fill_(1.5))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print("%s: %.3f ms  %.2f TB/s" % (name, dt * 1e3, x.numel() * 8 / dt / 1e12))
y = torch.empty_like(x)
y.copy_(x); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("copy: %.3f ms  %.2f TB/s (read+write)" % (dt * 1e3, 2 * x.numel() * 8 / dt / 1e12))
