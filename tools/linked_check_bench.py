"""`rawcooked_linked --check` (INTEGRATION.md route C) on N 4K frames, timed: bench.py's check.linked_check record on its own.
  python tools/linked_check_bench.py [frames] [cpu_frames] [variant,variant]      (RCGPU_* are inherited by the runs)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rawcooked_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 48
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else None
W, H, F = 4096, 2160, 16
frames = bench.make_frames(torch, F, W, H, "film", 1, "cuda:0")
line_bytes = W * 6
enc = api.Ffv1Encoder(W, H, synth.PIX_RGB16_BE, line_bytes, 8, 8, 1, 1, max_batch=F)
stride = (enc.max_packet + 255) & ~255
d_packets = torch.empty(F * stride, dtype=torch.uint8, device="cuda:0")
d_sizes = torch.zeros(F, dtype=torch.int64, device="cuda:0")
enc.encode_device([frames[i].data_ptr() for i in range(F)], d_packets.data_ptr(), stride, d_sizes.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
sizes = d_sizes.cpu().tolist()
record = enc.config_record()
enc.close()
print(json.dumps(bench.linked_check_record(api, synth, record, frames, d_packets, stride, sizes, W, H, synth.PIX_RGB16_BE, nframes=n, cpu_frames=m, variants=variants), indent=1))
