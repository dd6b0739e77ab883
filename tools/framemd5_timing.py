import sys, time, hashlib
sys.path.insert(0, '.')
import torch
from rawcooked_amd import api, synth
w, h, n = 4096, 2160, 16
comp = synth.components(w, h, 3, 16, "film", seed=1)
payload, lb = synth.pack_payload(comp, synth.PIX_RGB16_BE)
enc = api.Ffv1Encoder(w, h, synth.PIX_RGB16_BE, lb, 8, 8, max_batch=n)
t0 = time.time(); enc.encode_host([payload] * n); t1 = time.time()
sums, fb = enc.framemd5_last(n); t2 = time.time()
print("encode_host %d frames %.3f s, framemd5_last %.3f s, frame_bytes %d, ok=%s" % (n, t1 - t0, t2 - t1, fb, sums == [hashlib.md5(payload).digest()] * n))
enc.close()
