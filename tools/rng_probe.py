import torch, time
R, V = 256, 626
def t(n=30):
    t0 = time.perf_counter()
    for _ in range(n): torch.empty(R, V).exponential_(1)
    return (time.perf_counter() - t0) / n * 1e3
print("threads", torch.get_num_threads(), "exponential_ ms/call", round(t(), 3))
x = torch.empty(R, V); p = torch.empty(32, R, V).pin_memory()
t0 = time.perf_counter()
for j in range(32): p[j].copy_(x)
print("32 pinned copies ms", round((time.perf_counter() - t0) * 1e3, 3))
for n in (1, 8, 32):
    torch.set_num_threads(n); print("threads", n, "ms/call", round(t(), 3))
