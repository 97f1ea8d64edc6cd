"""Does a second stream's H2D copy run at full link rate beside the first one's?  Page-locked 39 MB pieces (a 128-frame VGA sub-batch),
16 pieces, on one stream and dealt over two; with and without a compute kernel stream busy beside them."""
import time, torch
n, piece = 16, 128 * 640 * 480
host = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(n)]
dev = [torch.empty(piece, dtype=torch.uint8, device="cuda") for _ in range(n)]
s = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
big = torch.randn(4096, 4096, device="cuda")
def run(nstreams, busy):
    torch.cuda.synchronize()
    t = time.perf_counter()
    if busy:
        with torch.cuda.stream(s[2]):
            for _ in range(40):
                big @ big
    for i in range(n):
        with torch.cuda.stream(s[i % nstreams]):
            dev[i].copy_(host[i], non_blocking=True)
    for k in range(nstreams):
        s[k].synchronize()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return n * piece / dt / 1e9
for busy in (False, True):
    for ns in (1, 2):
        run(ns, busy)
        print(f"{ns} stream(s), compute beside: {busy}: {max(run(ns, busy) for _ in range(3)):.1f} GB/s")
# H2D and D2H at the same time: is the link full duplex for two streams of one process?
hback = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(n)]
def duplex(d2h_share):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(s[0]):
            dev[i].copy_(host[i], non_blocking=True)
        if i % d2h_share == 0:
            with torch.cuda.stream(s[1]):
                hback[i].copy_(dev[(i + 8) % n], non_blocking=True)
    s[0].synchronize(); s[1].synchronize()
    dt = time.perf_counter() - t
    return n * piece / dt / 1e9, (n // d2h_share) * piece / dt / 1e9
for share in (1, 5):
    duplex(share)
    r = max(duplex(share) for _ in range(3))
    print(f"H2D {r[0]:.1f} GB/s with D2H {r[1]:.1f} GB/s beside it (one D2H piece per {share} H2D pieces)")
