import time, torch
n, world = 4_000_000_000, 8
recv = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    t = time.time(); part = torch.sum(recv.view(world, n // world), dim=0, dtype=torch.int32); torch.cuda.synchronize()
    print("torch.sum: %.1f ms, peak mem %.1f GB" % ((time.time() - t) * 1e3, torch.cuda.max_memory_allocated() / 1e9))
ref = recv[: n // world].clone()
t = time.time()
for r in range(1, world):
    ref += recv[r * (n // world):(r + 1) * (n // world)]
torch.cuda.synchronize()
print("in-place adds: %.1f ms; equal: %s" % ((time.time() - t) * 1e3, bool((ref == part).all())))
