"""Diagnostic: pinned host -> HBM copy rate of 236 MB blocks (one 256-stream ring entry at 1280x720) - one stream, four
streams at once, and beside a GPU that is busy (a stand-in compute loop) - to see what bounds the ingest-inclusive rate."""
import time
import torch
N = 256 * 720 * 1280
dev = torch.device("cuda", 0)
src = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
busy = torch.randn(8192, 8192, device=dev)

def run(nstreams, with_compute):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for i in range(4):
            with torch.cuda.stream(streams[i % nstreams]):
                dst[i].copy_(src[i], non_blocking=True)
        if with_compute:
            for _ in range(6):
                busy @ busy
    for s in streams:
        s.synchronize()
    t_copy = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 12 * N / t_copy / 1e9

for ns in (1, 2, 4):
    for wc in (False, True):
        r = run(ns, wc); r = run(ns, wc)
        print(f"{ns} copy stream(s), compute {'on ' if wc else 'off'}: {r:.1f} GB/s", flush=True)
