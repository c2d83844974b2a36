"""Host cost of allocating the lookup's output tensor, a few ways."""
import time, torch
dev = torch.device("cuda", 0)
shape = (1, 324, 55, 128)
c = torch.zeros(1, 2, 55, 128, device=dev)
tmpl = torch.empty(shape, device=dev)
def t(fn, n=50000):
    for _ in range(1000): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print(f"torch.empty(shape, dtype=, device=)  {t(lambda: torch.empty(shape, dtype=torch.float32, device=dev)):.2f} us")
print(f"c.new_empty(shape)                   {t(lambda: c.new_empty(shape)):.2f} us")
print(f"torch.empty_like(template)           {t(lambda: torch.empty_like(tmpl)):.2f} us")
e = torch._C._VariableFunctions.empty
print(f"_VariableFunctions.empty             {t(lambda: e(shape, dtype=torch.float32, device=dev)):.2f} us")
