"""FLAC leg of config 3 (6 ch / 24 bit / 48 kHz, 41.7 s = 1000 frames at 24 fps): encode on the device, compare the first blocks with the
oracle, print the call time.  Under rocprofv3 (tools/profile_flac.sh) this gives k_flac_plan / k_flac_write their kernel times."""
import json
import os
import sys
import time

import torch  # noqa: F401  (its HIP runtime first, see tests/conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rawcooked_amd import api, synth  # noqa: E402

ch, bits, rate = 6, 24, 48000
n = int(os.environ.get("FLAC_SAMPLES", str(rate * 1000 // 24)))
pcm = synth.pcm_samples(n, ch, bits, rate, "music", seed=7)
raw = synth.wav_file(pcm, bits, rate)[44:][: n * ch * bits // 8]
enc = api.FlacEncoder(ch, rate, bits)
enc.encode(raw[: 4096 * ch * 3])                      # warm-up: module load, buffers
t0 = time.perf_counter()
frames, cp = enc.encode(raw)
dt = time.perf_counter() - t0
import oracle_binding as ob  # noqa: E402
k = 8
oframes = ob.flac_encode(ch, rate, bits, raw[: k * 4608 * ch * 3])[0]
same = all(frames[i] == oframes[i] for i in range(min(k - 1, len(oframes) - 1)))
print(json.dumps({"samples_per_channel": n, "blocks": len(frames), "seconds": round(dt, 4), "realtime_factor": round(n / rate / dt, 1),
       "pcm_MB": round(len(raw) / 1e6, 1), "flac_MB": round(sum(map(len, frames)) / 1e6, 1), "first_blocks_equal_oracle": same}))
