"""How long does a chain of tiny dependent kernels on one stream take while another stream keeps the chip full of a wide kernel's
workgroups — with equal stream priorities and with the tiny kernels' stream at high priority?  (Why the two contexts of the measured
loop fall into lockstep: DESIGN.md §5.)  Measurement aid; torch kernels stand in for the product's."""
import time, torch
dev = torch.device("cuda:0")
big = torch.randint(0, 2**31 - 1, (1 << 28,), dtype=torch.int32, device=dev)   # 1 GiB: an elementwise pass over it is a ~0.5 ms wide kernel
small = torch.zeros(64, dtype=torch.int32, device=dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (least, greatest):", lo, hi)

def chain(stream, n=200):
    with torch.cuda.stream(stream):
        for _ in range(n):
            small.add_(1)

def wide(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            big.mul_(3)
            big.bitwise_xor_(0x5bd1e995)

for label, prio_small in (("equal priorities", 0), ("chain stream at high priority", hi)):
    s_wide = torch.cuda.Stream(priority=0)
    s_small = torch.cuda.Stream(priority=prio_small)
    torch.cuda.synchronize()
    chain(s_small); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(s_small); s_small.synchronize(); t_alone = time.perf_counter() - t0
    wide(s_wide, 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); wide(s_wide, 40); s_wide.synchronize(); t_wide_alone = time.perf_counter() - t0
    wide(s_wide, 40)
    time.sleep(0.002)
    t0 = time.perf_counter(); chain(s_small); s_small.synchronize(); t_busy = time.perf_counter() - t0
    s_wide.synchronize(); t_both = time.perf_counter() - t0
    print(f"{label}: chain of 200 tiny kernels alone {1e3 * t_alone:.2f} ms; while the wide stream runs {1e3 * t_busy:.2f} ms "
          f"(wide stream alone {1e3 * t_wide_alone:.2f} ms for 80 kernels)")
