"""what a plain streaming read of the config-3 stage matrix (107 GB) costs on this box: torch reductions / copies as the yardstick
for the HBM-bound kernels (rows_dot, the expression evaluator)"""
import torch, time
n = (1 << 27) * 100
x = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device="cuda")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / reps
t = timeit(lambda: x.sum()); print("int64 sum of 107 GB: %.2f ms, %.2f TB/s" % (t * 1e3, n * 8 / t / 1e12), flush=True)
t = timeit(lambda: x.view(torch.float64).sum()); print("f64 sum of 107 GB: %.2f ms, %.2f TB/s" % (t * 1e3, n * 8 / t / 1e12), flush=True)
t = timeit(lambda: (x.view(torch.int32).max())); print("int32 max of 107 GB: %.2f ms, %.2f TB/s" % (t * 1e3, n * 8 / t / 1e12), flush=True)
y = torch.empty(n // 2, dtype=torch.int64, device="cuda")
t = timeit(lambda: y.copy_(x[:n // 2])); print("copy 53.7 GB -> 53.7 GB: %.2f ms, %.2f TB/s (read+write)" % (t * 1e3, n * 8 / t / 1e12), flush=True)
