"""Is the intermittent 60-90 ms stall of `CodecEngine.to_host` the HOST copy out of the pinned staging buffer (torch's OpenMP pool
waking up on a many-core / quota-limited host)?  Times the clone alone, after a GPU-side wait like the real loop, by method."""
import os, time
import numpy as np
import torch
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)),
      "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", flush=True)
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)

def stats(name, ts):
    ts = np.array(ts) * 1e3
    print(f"{name}: median {np.median(ts):.2f} ms, max {ts.max():.1f} ms, > 30 ms: {(ts > 30).sum()}/{len(ts)}", flush=True)

for n_el, label, reps in ((16 * 12000, "768 KB", 40), (64 * 261888, "67 MB", 10)):
    pin = torch.empty(n_el, dtype=torch.float32).pin_memory()
    methods = {
        "torch clone (at::parallel_for)": lambda: pin.clone().numpy(),
        "numpy copy (one thread)": lambda: np.array(pin.numpy(), copy=True),
    }
    for name, fn in methods.items():
        ts = []
        for _ in range(reps):
            for _ in range(6):
                y = x @ x          # ~7 ms of GPU work the host waits for, like a window decode
            ev = torch.cuda.Event(); ev.record()
            while not ev.query():
                time.sleep(0)
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        stats(f"[{label}] {name}", ts)
torch.set_num_threads(8)
pin = torch.empty(16 * 12000).pin_memory(); ts = []
for _ in range(40):
    for _ in range(6):
        y = x @ x
    torch.cuda.synchronize(); t0 = time.perf_counter(); pin.clone(); ts.append(time.perf_counter() - t0)
stats("[768 KB] torch clone with torch.set_num_threads(8)", ts)
