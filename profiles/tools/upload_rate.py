"""Diagnostic: host-call time and achieved rate of mvo_batch_upload_async (pinned host -> device ring), alone and beside a step."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ros2_mono_vo_amd import Context
W, H, B = 1280, 720, 256
with Context(max_width=W, max_height=H, batch=B, nfeatures=2000, max_points=4096, ring_frames=2) as ctx:
    pin = ctx.host_alloc(B * H * W).reshape(B, H, W)
    pin[:] = 7
    for rep in range(4):
        t0 = time.perf_counter()
        ctx.batch_upload_async(rep % 2, pin.ctypes.data, W, H, W, H * W)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        print(f"upload {B*H*W/1e6:.0f} MB: host call {1e3*(t1-t0):.2f} ms, complete after {1e3*(t2-t0):.2f} ms = {B*H*W/(t2-t0)/1e9:.1f} GB/s", flush=True)
    pag = np.full((B, H, W), 9, np.uint8)
    t0 = time.perf_counter()
    ctx.batch_upload_async(0, pag.ctypes.data, W, H, W, H * W)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"pageable: host call {1e3*(t1-t0):.2f} ms, complete after {1e3*(t2-t0):.2f} ms = {B*H*W/(t2-t0)/1e9:.1f} GB/s")
