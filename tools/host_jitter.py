"""Where do rare 20-40 ms host stalls come from?  Max gap between consecutive perf_counter reads in a tight Python loop
(no GPU work), then while issuing tiny GPU launches; plus the cgroup's CPU throttling counters before / after."""
import gc
import os
import time


def cg(name):
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
        p = os.path.join(base, name)
        if os.path.exists(p):
            return open(p).read().strip().replace("\n", " ")
    return "n/a"


def gaps(seconds, work=None):
    t_end = time.perf_counter() + seconds
    last = time.perf_counter()
    worst, n, over = 0.0, 0, []
    while last < t_end:
        if work:
            work()
        now = time.perf_counter()
        d = now - last
        if d > 0.005:
            over.append((round((now - (t_end - seconds)), 3), round(d * 1e3, 1)))
        worst = max(worst, d)
        last = now
        n += 1
    return worst * 1e3, n, over


gc.disable()
print("cpu.max:", cg("cpu.max"), "| cpu.stat:", cg("cpu.stat"))
print("affinity:", len(os.sched_getaffinity(0)), "cpus; loadavg", os.getloadavg())
w, n, over = gaps(4.0)
print(f"python only: {n} iterations, worst gap {w:.2f} ms, gaps > 5 ms (at s, ms): {over}")
import torch

x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
w, n, over = gaps(6.0, lambda: x.add_(1.0))
torch.cuda.synchronize()
print(f"tiny launches: {n} launches, worst gap {w:.2f} ms, gaps > 5 ms (at s, ms): {over}")
w, n, over = gaps(4.0, lambda: (x.add_(1.0), torch.cuda.synchronize()))
print(f"launch + sync: {n} iterations, worst gap {w:.2f} ms, gaps > 5 ms (at s, ms): {over}")
print("cpu.stat after:", cg("cpu.stat"))
