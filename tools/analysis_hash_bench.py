"""Route D in numbers: `rawcooked_linked -d` (the reference's analysis pass, patched by oracle/route_d_*.patch) over N 4K DPX files on
tmpfs, with the whole-file MD5s and the padding-bit test from the device (rcgpu_analysis_host_batch, files side by side) and from the
reference's own loops (RCGPU_HASH=0).
Usage on the GPU box: python tools/analysis_hash_bench.py [files] [16|10]       (10: RGB 10-bit FilledA, analysed with --check-padding)"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rawcooked_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h = 4096, 2160
work = "/dev/shm/rcgpu_rd_%d" % os.getpid()
os.makedirs(work + "/seq")
try:
    rng = np.random.default_rng(1)
    if bits == 16:
        pixfmt, args = synth.PIX_RGB16_BE, ["--hash", "--no-check-padding"]
        pl = rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16).tobytes()
    else:
        pixfmt, args = synth.PIX_RGB10_FILLEDA_BE, ["--hash", "--check-padding"]
        pl = (rng.integers(0, 1 << 30, size=(h, w), dtype=np.uint32) << 2).astype(">u4").tobytes()       # three 10-bit samples, two zero filler bits
    for i in range(n):
        with open(work + "/seq/f_%06d.dpx" % i, "wb") as f:
            f.write(synth.dpx_file(None, pixfmt, frame_index=i, payload=pl, size=(w, h)))
    exe = os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")
    out = {}
    for name, env in (("device", {"RCGPU_HASH": "1"}), ("reference", {"RCGPU_HASH": "0"}), ("device", {"RCGPU_HASH": "1"})):
        t0 = time.perf_counter()
        r = subprocess.run([exe] + args + ["-d", "-y", "seq"], cwd=work, capture_output=True, text=True, env=dict(os.environ, **env), timeout=600, stdin=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        data = open(work + "/seq.rawcooked_reversibility_data", "rb").read()
        out.setdefault(name, []).append((dt, data))
        print("%-9s rc %d  %.2f s for %d 4K %d-bit files (%s) = %.1f files/s = %.2f GB/s" % (name, r.returncode, dt, n, bits, " ".join(args), n / dt, n * len(pl) / dt / 1e9))
    print("same reversibility data:", out["device"][0][1] == out["reference"][0][1] == out["device"][1][1])
finally:
    shutil.rmtree(work, ignore_errors=True)
