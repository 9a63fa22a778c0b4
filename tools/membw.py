import torch, time
dev="cuda:0"
n = 1<<30
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e-3
dt = t(lambda: a.copy_(b)); print(f"copy 1GiB: {dt*1e6:.0f} us  read+write {2*n/dt/1e12:.2f} TB/s")
dt = t(lambda: a.zero_()); print(f"memset 1GiB: {dt*1e6:.0f} us  write {n/dt/1e12:.2f} TB/s")
x = torch.empty(n//4, dtype=torch.float32, device=dev)
dt = t(lambda: x.sum()); print(f"sum 1GiB: {dt*1e6:.0f} us  read {n/dt/1e12:.2f} TB/s")
m = 32<<20
a2 = torch.empty(m, dtype=torch.uint8, device=dev); b2 = torch.empty(m, dtype=torch.uint8, device=dev)
dt = t(lambda: a2.copy_(b2), 200); print(f"copy 32MiB: {dt*1e6:.1f} us  read+write {2*m/dt/1e12:.2f} TB/s")
dt = t(lambda: a2.zero_(), 200); print(f"memset 32MiB: {dt*1e6:.1f} us  write {m/dt/1e12:.2f} TB/s")
