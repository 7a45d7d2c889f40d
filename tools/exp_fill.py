import torch, time
x = torch.empty(256*1024*1024, device="cuda")   # 1 GiB
for name, fn in (("zero_", lambda: x.zero_()), ("fill_", lambda: x.fill_(1.5)), ("copy_", None)):
    if fn is None:
        y = torch.empty_like(x); fn = lambda: y.copy_(x)
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    nbytes = x.numel() * 4 * (2 if name == "copy_" else 1)
    print(name, f"{dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s")
# strided row pattern like the conv epilogue: write [M][HW] rows where each wave instruction writes 128 B of 2 rows
