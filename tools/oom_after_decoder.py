"""Scratch (round 5): which use of the check-half decoder makes a LATER out-of-memory hipMalloc in the same process segfault inside the HIP runtime?
   python tools/oom_after_decoder.py <variant>      none | create | create_alive | keep | verify | hint | hint_fail"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NO_TORCH = os.environ.get("NO_TORCH") == "1"      # without torch the process has /opt/rocm's HIP runtime, with it the one torch bundles
if not NO_TORCH:
    import torch
from rawcooked_amd import api, synth

v = sys.argv[1]
w, h, pixfmt, nh, nv = 192, 96, synth.PIX_RGB16_BE, 2, 2
srcs = [synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=300 + i), pixfmt, True) for i in range(3)]
line_bytes = srcs[0][1]
pls = [s[0] for s in srcs]
enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3)
packets = enc.encode_host(pls)
enc.close()
if v != "none":
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3)
    if v in ("keep", "verify", "hint", "hint_fail"):
        dec.decode_keep(packets)
    if v == "verify":
        dec.verify_kept([{"slot": i, "before": b"", "after": b"", "on_disk": pls[i]} for i in range(3)])
    if v in ("hint", "hint_fail"):
        path = "/tmp/blocks.bin"
        offs, pos, blob = [], 11, bytearray(11)
        for p_ in packets:
            offs.append(pos); blob += p_ + b"\0" * 7; pos += len(p_) + 7
        open(path, "wb").write(blob)
        if v == "hint_fail":
            offs[0] += 3
        dec.decode_keep_hint_file(path, offs, [len(p_) for p_ in packets])
        try:
            dec.decode_keep_adopt()
        except api.RcgpuError as e:
            print("adopt:", str(e)[:80])
    if v != "create_alive":
        dec.close()
hip = ctypes.CDLL("libamdhip64.so")
if NO_TORCH:
    fb, tb = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipDeviceSynchronize(); hip.hipMemGetInfo(ctypes.byref(fb), ctypes.byref(tb))
    f_ = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(f_), ctypes.c_size_t(fb.value - (96 << 20))) == 0
else:
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    filler = torch.empty(free0 - (96 << 20), dtype=torch.uint8, device="cuda")
p = ctypes.c_void_p()
rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 30))
print("variant", v, "torch" if not NO_TORCH else "no torch", ": hipMalloc of 1 GiB with 96 MiB free returned", rc)
