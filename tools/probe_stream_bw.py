"""Practical HBM streaming ceiling of the box, for calibrating roofline fractions: read-only (sum), copy and fill of buffers far
larger than the 256 MiB Infinity Cache, HIP-event timed."""
import torch
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
for mib in (128, 512, 2048):
    n = mib * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    t_sum = timed(lambda: x.sum()); t_copy = timed(lambda: y.copy_(x)); t_fill = timed(lambda: y.fill_(1.0)); t_add = timed(lambda: torch.add(x, 1.0, out=y))
    b = n * 4
    print(f"{mib:5d} MiB  read(sum) {b / t_sum / 1e12:5.2f} TB/s   copy {2 * b / t_copy / 1e12:5.2f} TB/s   fill {b / t_fill / 1e12:5.2f} TB/s   add(r+w) {2 * b / t_add / 1e12:5.2f} TB/s")
