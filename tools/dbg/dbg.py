import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import oracle_binding as ob
from rawcooked_amd import api, synth
w, h, pixfmt, nh, nv, nframes = 200, 120, synth.PIX_RGB16_BE, 3, 2, 3
for segments in (1, 0, 2):
  for kinds in (["noise","film","film"], ["film","film","film"], ["noise"]):
    payloads = []
    for i,k in enumerate(kinds):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, k, seed=50 + i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=len(kinds), segments=segments)
    for rep in range(2):
        try:
            packets = enc.encode_host(payloads)
            ok = [packets[f] == ob.encode_payload(p, payloads[f], line_bytes) for f in range(len(kinds))]
            print(segments, kinds, rep, ok, [len(x) for x in packets], len(payloads[0]))
        except Exception as ex:
            print(segments, kinds, rep, 'EXC', str(ex)[-80:], [len(ob.encode_payload(p, pl, line_bytes)) for pl in payloads])
    enc.close()
