import time, torch
dev = torch.device("cuda", 0)
n = 64 * 1024 * 1024
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device=dev)
d2 = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1): d1.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2): h_out.copy_(d2, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return reps * n / dt / 1e9
run(True, True, 2)
print("H2D alone  %.1f GB/s" % run(True, False))
print("D2H alone  %.1f GB/s" % run(False, True))
print("both       %.1f GB/s each direction" % run(True, True))
