#!/usr/bin/env python3
"""Facts about the GPU box's host side that size the upload/download pipeline: cores, memory, tmpfs, PCIe rates
(pinned and pageable, one direction and both at once), and the cost of page-locking."""
import os, subprocess, time, sys
import torch

def sh(c):
    try: return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e: return "ERR %s" % e

print("nproc", os.cpu_count())
print(sh("free -g | head -3"))
print(sh("df -h /dev/shm /tmp . | cat"))
print(sh("lscpu | egrep 'Model name|Socket|NUMA node|Thread' | cat"))
print(sh("numactl -H 2>/dev/null | head -20"))
dev = torch.device("cuda", 0)
GB = 1 << 30
n = 4 * GB
t0 = time.perf_counter(); hp = torch.empty(n, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
print("pin alloc 4 GiB: %.3f s (%.2f GB/s)" % (t1 - t0, n / (t1 - t0) / 1e9))
t0 = time.perf_counter(); hp.fill_(1); t1 = time.perf_counter(); print("fill pinned 4 GiB (1 thread): %.2f GB/s" % (n / (t1 - t0) / 1e9))
hp2 = torch.empty(n, dtype=torch.uint8, pin_memory=True); hp2.fill_(2)
d = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def tm(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print("H2D pinned: %.2f GB/s" % (n / tm(lambda: d.copy_(hp, non_blocking=True)) / 1e9))
print("D2H pinned: %.2f GB/s" % (n / tm(lambda: hp2.copy_(d2, non_blocking=True)) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): hp2.copy_(d2, non_blocking=True)
t = tm(both); print("both at once: %.2f GB/s each way (%.3f s for 2 x 4 GiB)" % (n / t / 1e9, t))
pg = torch.empty(n, dtype=torch.uint8); pg.fill_(3)
print("H2D pageable: %.2f GB/s" % (n / tm(lambda: d.copy_(pg), 2) / 1e9))
print("D2H pageable: %.2f GB/s" % (n / tm(lambda: pg.copy_(d2), 2) / 1e9))
# host memcpy rate with several threads (readers filling a pinned ring from the page cache)
import threading, numpy as np
src = pg.numpy(); dst = hp.numpy()
for nt in (1, 4, 8, 16):
    def work(i):
        a = i * (n // nt); b = a + n // nt
        np.copyto(dst[a:b], src[a:b])
    t0 = time.perf_counter(); th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    [x.start() for x in th]; [x.join() for x in th]; t1 = time.perf_counter()
    print("host memcpy pageable->pinned, %d threads: %.2f GB/s" % (nt, n / (t1 - t0) / 1e9))
# tmpfs write/read
p = "/dev/shm/rcgpu_probe.bin"
t0 = time.perf_counter(); open(p, "wb").write(memoryview(src)[:2 * GB]); t1 = time.perf_counter()
print("tmpfs write 2 GiB: %.2f GB/s" % (2 * GB / (t1 - t0) / 1e9))
t0 = time.perf_counter(); f = open(p, "rb"); f.readinto(memoryview(dst)[:2 * GB]); t1 = time.perf_counter()
print("tmpfs read into pinned 2 GiB: %.2f GB/s" % (2 * GB / (t1 - t0) / 1e9))
os.unlink(p)
print(torch.cuda.get_device_properties(0))
print("mem_get_info", torch.cuda.mem_get_info())
