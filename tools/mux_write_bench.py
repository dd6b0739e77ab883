#!/usr/bin/env python3
"""How fast does the muxer's parallel-writer path take packets?  N blocks of 49 MB through rcgpu_mkv_expect / reserve_block / prefault +
copy from T threads, on /dev/shm (mapped, pages allocated ahead) and on /tmp (pwrite).  No GPU involved.
    python tools/mux_write_bench.py [blocks] [threads]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rawcooked_amd import api   # noqa: E402
import numpy as np              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = 48847592
src = np.random.default_rng(1).integers(0, 255, size, dtype=np.uint8)
L = api.lib()
for base, env in (("/dev/shm", {}), ("/dev/shm", {"RCGPU_MKV_NO_MMAP": "1"}), ("/tmp", {})):
    for k, v in env.items():
        os.environ[k] = v
    path = os.path.join(base, "rcgpu_muxbench.mkv")
    mux = api.MkvMuxer(path)
    tv = mux.add_video(b"\x01\x02\x03", 4096, 2160, 24, 1)
    mux.begin()
    assert L.rcgpu_mkv_expect(mux.h, n * 96 * (1 << 20), n) == 0
    t0 = time.perf_counter()
    jobs = []
    for i in range(n):                       # the placer: all blocks laid out at once (what a finished batch does)
        dst, off = C.c_void_p(), C.c_uint64()
        assert L.rcgpu_mkv_reserve_block(mux.h, tv, i * 10 ** 9 // 24, size, 1, C.byref(dst), C.byref(off)) == 0
        jobs.append((dst.value, off.value))
    t_place = time.perf_counter() - t0
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= n:
                return
            dst, off = jobs[i]
            if dst:
                L.rcgpu_mkv_copy_in(mux.h, dst, src.ctypes.data, size)
            else:
                assert L.rcgpu_mkv_fill(mux.h, off, src.ctypes.data, size) == 0
    ths = [threading.Thread(target=work) for _ in range(T)]
    [t.start() for t in ths]; [t.join() for t in ths]
    t_copy = time.perf_counter() - t0
    mux.close()
    t_all = time.perf_counter() - t0
    print(f"{base:9s} {'mapped' if jobs[0][0] else 'pwrite'}: {n} blocks of {size >> 20} MiB, {T} threads: placed in {t_place:.3f} s, copied after {t_copy:.2f} s "
          f"({n * size / t_copy / 1e9:.2f} GB/s = {n / t_copy:.0f} blocks/s), closed after {t_all:.2f} s", flush=True)
    os.unlink(path)
    for k in env:
        del os.environ[k]
