"""How long the device decoder takes for a batch of N 4K frames (one lane per slice: the batch's time is a chain's latency).
  python tools/dec_latency.py [frames ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rawcooked_amd import api, synth  # noqa: E402

W, H, F = 4096, 2160, 16
frames = bench.make_frames(torch, F, W, H, "film", 1, "cuda:0")
enc = api.Ffv1Encoder(W, H, synth.PIX_RGB16_BE, W * 6, 8, 8, 1, 1, max_batch=F)
stride = (enc.max_packet + 255) & ~255
d_packets = torch.empty(F * stride, dtype=torch.uint8, device="cuda:0")
d_sizes = torch.zeros(F, dtype=torch.int64, device="cuda:0")
enc.encode_device([frames[i].data_ptr() for i in range(F)], d_packets.data_ptr(), stride, d_sizes.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
sizes = d_sizes.cpu().tolist()
enc.close()
for n in [int(a) for a in sys.argv[1:]] or [64, 256]:
    dec = api.Ffv1Decoder(W, H, synth.PIX_RGB16_BE, W * 6, 8, 8, 1, 1, max_batch=n)
    out = torch.empty((n, W * H * 6), dtype=torch.uint8, device="cuda:0")
    pk = [d_packets.data_ptr() + (i % F) * stride for i in range(n)]
    sz = [sizes[i % F] for i in range(n)]
    for rep in range(2):
        t0 = time.perf_counter()
        flags = dec.decode_device(pk, sz, [out[i].data_ptr() for i in range(n)])
        dt = time.perf_counter() - t0
    ok = all(bool(torch.equal(out[i], frames[i % F])) for i in range(0, n, max(1, n // 8)))
    print("%4d frames: %.3f s  (%.1f frames/s)  kernels %s  flags %d  payloads %s" % (n, dt, n / dt, {k: round(v) for k, v in dec.kernel_times().items()}, flags, "ok" if ok else "DIFFER"), flush=True)
    dec.close(); del out
