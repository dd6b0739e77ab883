"""SURVEY.md 8d's H2D-inclusive figure on BASELINE config 2's own 1000 frames, over (batch, lanes per device, run-on): one JSON row per setting.
Payloads start in PINNED host memory (a ring of 32 distinct frames) and packets end in pageable host memory, uploads / coding / downloads overlapped
(rcgpu_ffv1_encode_sequence_memory, frames_pinned = 1) -- what bench.py's host_pipeline_pinned_inputs_1000_frames record times with the default setting.
Every setting's packet sizes and the MD5 of its last packets must equal the first setting's (the bytes of a frame do not depend on how it was batched).
    python tools/h2d_sweep.py [--frames 1000] [--settings 336x1,168x2,...]  >  profiles/r06_h2d_1000.jsonl        (batchxlanes[r] : r = run-on encoders)"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--settings", default="336x1,336x1r,250x1,168x2,168x2r,112x3,125x2,84x4,200x1r")
    ap.add_argument("--rc-span", type=int, default=0, help="rcgpu_ffv1_config::rc_span of the encoders (0 automatic, 1 whole slices, >= 8 split coder)")
    ap.add_argument("--idle", type=float, default=4.0, help="seconds between two settings (the driver wipes freed memory in the background)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from rawcooked_amd import api, synth
    dev = torch.device("cuda", 0)
    W, H = bench.W4K, bench.H4K
    ring = bench.make_frames(torch, 32, W, H, "film", 0, dev)
    pins = [ring[i].cpu().pin_memory() for i in range(32)]
    del ring
    torch.cuda.empty_cache()
    addrs = [t.data_ptr() for t in pins]
    payload = pins[0].numel()
    nh, nv = api.slices_to_grid(args.slices)
    nout = 64
    out_cap = int(payload * 1.3) + (1 << 20)
    outs = [np.empty(out_cap, dtype=np.uint8) for _ in range(nout)]
    want = None
    for s in args.settings.split(","):
        run_on = s.endswith("r")
        b, l = (int(x) for x in s.rstrip("r").split("x"))
        cfg = api.Ffv1Config(W, H, synth.PIX_RGB16_BE, W * 6, nh, nv, 1, 1, 0, 0, 0, 0, 1, 3, args.rc_span, 0)
        time.sleep(args.idle)
        t0 = time.perf_counter()
        try:
            st, sizes = api.encode_sequence_memory(cfg, addrs, args.frames, [a.ctypes.data for a in outs], out_cap, batch=b, device_first=0, device_count=1, lanes_per_device=l,
                                                   frames_pinned=1, run_on=1 if run_on else 0)
        except Exception as e:
            print(json.dumps({"setting": s, "error": str(e)[-300:]})); sys.stdout.flush()
            continue
        wall = time.perf_counter() - t0
        sig = (sizes, [hashlib.md5(bytes(outs[f % nout][:sizes[f]])).hexdigest() for f in range(args.frames - 8, args.frames)])
        if want is None:
            want = sig
        print(json.dumps({"setting": s, "rc_span": args.rc_span, "ramp": bool(os.environ.get("RCGPU_RAMP")), "frames": args.frames, "batch": b, "lanes_per_device": l, "run_on": run_on, "frames_per_second": round(args.frames / st.seconds, 1),
                          "seconds": round(st.seconds, 3), "call_seconds": round(wall, 3), "prepare_seconds": round(st.prepare_seconds, 3), "first_packet_seconds": round(st.first_packet_seconds, 3),
                          "last_batch_seconds": round(st.last_batch_seconds, 3), "steady_frames_per_second": round(st.steady_frames_per_second, 1), "batches": st.batches,
                          "batch_frames": st.batch_frames, "packets_equal_to_first_setting": sig == want}))
        sys.stdout.flush()
    api.lib().rcgpu_release_device_streams()


if __name__ == "__main__":
    main()
