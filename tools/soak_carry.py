"""Soak for the rarest path of k_rangecode: a carry that ripples out of the second-stage dword `pd` (once per ~2^32 flushes, i.e. per
~17 GB of output) is handed to k_footer as an event and added to bytes already in HBM.  Noise frames make ~53 MB of output each, so
a few thousand 4K frames see a handful of events.  Every packet is decoded again by the device decoder (independent code) and
compared with its source.   GPU box:  python tools/soak_carry.py [frames]  """
import os, sys, struct
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rawcooked_amd import api, synth

def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
    W, H, B = 4096, 2160, int(os.environ.get("RCGPU_SOAK_BATCH", "64"))
    dev = torch.device("cuda:0")
    line_bytes = W * 6
    enc = api.Ffv1Encoder(W, H, synth.PIX_RGB16_BE, line_bytes, 8, 8, 1, 1, max_batch=B)
    dec = api.Ffv1Decoder(W, H, synth.PIX_RGB16_BE, line_bytes, 8, 8, 1, 1, max_batch=B)
    stride = (enc.max_packet + 255) & ~255
    packets = torch.empty(B * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(B, dtype=torch.int64, device=dev)
    outs = torch.empty((B, line_bytes * H), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(20260929)
    events = bad = done = 0
    while done < total:
        frames = torch.randint(0, 256, (B, line_bytes * H), dtype=torch.uint8, device=dev, generator=g)
        enc.encode_device([frames[i].data_ptr() for i in range(B)], packets.data_ptr(), stride, sizes.data_ptr())
        torch.cuda.synchronize()
        err, ev = struct.unpack("<II", enc.debug_fetch(5, 0, 16)[:8])
        assert err == 0, f"encoder error flags {err}"
        events += ev
        sz = sizes.cpu().tolist()
        flags = dec.decode_device([packets.data_ptr() + i * stride for i in range(B)], sz, [outs[i].data_ptr() for i in range(B)])
        torch.cuda.synchronize()
        if flags or not torch.equal(outs, frames):
            bad += 1
            print("MISMATCH in batch starting at frame", done, "decoder flags", flags, flush=True)
        done += B
    print(f"soak_carry: {done} frames, {events} carry events, {bad} bad batches")
    sys.exit(1 if bad else 0)

main()
