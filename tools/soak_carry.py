"""Full-size round-trip soak, written for the rarest path of k_rangecode: a carry that ripples out of the second-stage dword `pd` (once per ~2^32 flushes, i.e. per
~17 GB of output) is handed to k_footer as an event and added to bytes already in HBM.  Noise frames make ~53 MB of output each, so
a few thousand 4K frames see a handful of events.  Every packet is decoded again by the device decoder (independent code) and
compared with its source.   GPU box:  python tools/soak_carry.py [frames]  """
import os, sys, struct
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rawcooked_amd import api, synth

def mixed_frames(torch, g, dev, B, W, H):
    """16-bit RGB big-endian payloads: a smooth ramp per channel, grain whose amplitude differs per frame (0 .. 12 bits), and a few
    constant rectangles."""
    out = torch.empty((B, H, W, 3), dtype=torch.int32, device=dev)
    yy = torch.arange(H, device=dev, dtype=torch.int32).view(1, H, 1, 1)
    xx = torch.arange(W, device=dev, dtype=torch.int32).view(1, 1, W, 1)
    ch = torch.arange(3, device=dev, dtype=torch.int32).view(1, 1, 1, 3)
    for b0 in range(0, B, 16):
        n = min(16, B - b0)
        amp = torch.randint(0, 13, (n, 1, 1, 1), device=dev, generator=g, dtype=torch.int32)
        grain = (torch.randint(0, 1 << 30, (n, H, W, 3), device=dev, generator=g, dtype=torch.int32) & ((1 << amp) - 1))
        base = (xx * 7 + yy * 5 + ch * 9000 + 3000)
        out[b0:b0 + n] = (base + grain) & 0xFFFF
    for _ in range(6):
        y0 = int(torch.randint(0, H - 64, (1,), generator=g, device=dev)); x0 = int(torch.randint(0, W - 64, (1,), generator=g, device=dev))
        h = int(torch.randint(16, 600, (1,), generator=g, device=dev)); w = int(torch.randint(16, 1500, (1,), generator=g, device=dev))
        out[:, y0:y0 + h, x0:x0 + w, :] = int(torch.randint(0, 65536, (1,), generator=g, device=dev))
    hi = (out >> 8).to(torch.uint8); lo = (out & 0xFF).to(torch.uint8)
    return torch.stack((hi, lo), dim=-1).reshape(B, H * W * 6).contiguous()


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
        if os.environ.get("RCGPU_SOAK_KIND", "noise") == "noise":
            frames = torch.randint(0, 256, (B, line_bytes * H), dtype=torch.uint8, device=dev, generator=g)
        else:       # "mixed": ramps + grain of random depth + flat rectangles (runs of zero residuals, many lanes per context)
            frames = mixed_frames(torch, g, dev, B, W, H)
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
