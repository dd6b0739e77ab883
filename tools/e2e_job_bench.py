#!/usr/bin/env python3
"""End-to-end job timing on the GPU box: files on disk -> rawcooked analysis -> rcgpu-ffmpeg (GPU) -> MKV -> rawcooked --check.
Config 3 shape: 4K-DCI 16-bit DPX + 6 ch / 24 bit / 48 kHz WAV.   python tools/e2e_job_bench.py [frames]"""
import os
import shlex
import shutil
import subprocess
import sys
import time


def run(cmd, cwd, env=None, timeout=180, attempts=3):
    """The reference occasionally dead-locks in its own thread pool on many-core hosts: bounded, repeated (every step is idempotent)."""
    for a in range(attempts):
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, env=env, stdin=subprocess.DEVNULL, timeout=timeout)
        except subprocess.TimeoutExpired:
            print('step timed out, repeating:', ' '.join(cmd[:4]), file=sys.stderr)
            if a + 1 == attempts:
                raise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from rawcooked_amd import synth   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
work = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/rcgpu_e2e"
shutil.rmtree(work, ignore_errors=True)
os.makedirs(work + "/pkg/img")
W, H = 4096, 2160
frames = bench.make_frames(torch, n, W, H, "film", 0, torch.device("cuda", 0)).cpu().numpy()
hdr = synth.dpx_file(np.zeros((1, 1, 3), dtype=np.uint16), synth.PIX_RGB16_BE)[:2048]
for i in range(n):
    import struct
    h = bytearray(hdr)
    struct.pack_into(">I", h, 772, W); struct.pack_into(">I", h, 776, H); struct.pack_into(">I", h, 16, 2048 + frames.shape[1])
    struct.pack_into(">I", h, 1712, i)
    with open(f"{work}/pkg/img/f_{i:06d}.dpx", "wb") as f:
        f.write(h); f.write(frames[i].tobytes())
with open(work + "/pkg/snd.wav", "wb") as f:
    f.write(synth.wav_file(synth.pcm_samples(n * 2000, 6, 24, 48000), 24, 48000))
ref = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
t0 = time.time(); r = run([ref, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work); t_an = time.time() - t0
assert r.returncode == 0, r.stderr
argv = shlex.split(r.stdout.strip())
t0 = time.time(); r = run([shim] + argv[1:], work, env=dict(os.environ, RCGPU_TRACE="1")); t_enc = time.time() - t0
print(r.stderr)
assert r.returncode == 0, r.stdout + r.stderr
size = os.path.getsize(work + "/pkg.mkv")
t0 = time.time(); r = run([ref, "--check", "pkg.mkv"], work, timeout=600); t_chk = time.time() - t0
ok = r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout
print({"frames": n, "analysis_s(reference, 1 thread, MD5 of every file)": round(t_an, 2), "rcgpu_encode_s(files->mkv, incl. process start + device init)": round(t_enc, 2),
       "encode_fps_end_to_end": round(n / t_enc, 2), "mkv_bytes": size, "source_bytes": n * (2048 + frames.shape[1]),
       "reference_check_s(CPU decoder)": round(t_chk, 2), "reference_check_ok": ok})
shutil.rmtree(work, ignore_errors=True)
