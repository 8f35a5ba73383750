"""Stage times of ONE context stepping alone (no other context beside it): `python3 profiles/tools/kf_step_stages.py [B] [policy] [steps]`.
policy 1 = key-frame branch on every frame of every stream, 2 = LK + PnP only, 0 = the reference's policy.  Prints the mean
milliseconds per step of every stage timer (HIP events on the context's stream), i.e. the chip time of each stage when nothing
overlaps it - the number the mixed run of bench.py cannot give."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ros2_mono_vo_amd import Context, synth, synth_gpu

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
policy = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
W, H, NF = int(os.environ.get("KF_W", 1280)), int(os.environ.get("KF_H", 720)), int(os.environ.get("KF_NF", 2000))
NFR = steps + 4
dev = torch.device("cuda", 0)
Kmat = synth.default_K(W, H)
bank = synth_gpu.SceneBank(dev)
pitch = (W + 63) // 64 * 64
frames = torch.zeros((NFR, B, H, pitch), dtype=torch.uint8, device=dev)
depth0 = np.zeros((B, H, W), np.float32)
for s in range(B):
    fr, d0 = synth_gpu.render_stream(bank, bank.stream_params(0x5EED0003 + s), Kmat, W, H, NFR)
    frames[:, s, :, :W] = fr
    depth0[s] = d0.cpu().numpy()
torch.cuda.synchronize()
with Context(max_width=W, max_height=H, batch=B, nfeatures=NF, max_points=8192 if NF > 2000 else 4096, ring_frames=NFR, device=0) as ctx:
    ctx.batch_set_intrinsics(Kmat, np.zeros(5))
    for f in range(NFR):
        ctx.batch_upload_async(f, frames[f].data_ptr(), W, H, pitch, H * pitch)
    ctx.sync()
    ctx.batch_seed(0)
    for s in range(B):
        ctx.batch_set_landmarks(s, synth_gpu.depth_landmarks(Kmat, depth0[s], ctx.batch_get_tracks(s)))
    ctx.batch_set_policy(policy)
    for k in (1, 2):
        ctx.batch_track(k)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter()
    kf = 0
    for k in range(3, 3 + steps):
        kf += sum(1 for r in ctx.batch_track(k) if r.n_keypoints > 0)
    dt = (time.perf_counter() - t0) / steps * 1e3
    out = {"batch": B, "policy": policy, "steps": steps, "ms_per_step_wall": round(dt, 3), "keyframes_per_step": kf / steps}
    tot = 0.0
    for name in bench.STAGE_TIMERS:
        a, b = ctx.profile_read(name)
        if b:
            out[name] = round(a / steps, 4)
            if name not in ("pnp",):
                tot += a / steps
    out["sum_of_stages"] = round(tot, 3)
print(json.dumps(out))
