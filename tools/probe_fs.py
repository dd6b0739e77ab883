#!/usr/bin/env python3
"""How fast can ONE output file take bytes on the GPU box?  pwrite vs. a shared mapping, 1..16 threads, tmpfs and the local disk; and
how fast do hard-linked input files read with pread.  (Sizes the e2e record: the Matroska file is one inode.)"""
import ctypes, mmap, os, sys, threading, time
import numpy as np

GB = 1 << 30
def sh(c):
    import subprocess
    return subprocess.run(c, shell=True, capture_output=True, text=True).stdout.strip()
print("shmem_enabled:", sh("cat /sys/kernel/mm/transparent_hugepage/shmem_enabled"), "| thp:", sh("cat /sys/kernel/mm/transparent_hugepage/enabled"))
print(sh("uname -r"), "|", sh("mount | egrep ' /dev/shm | / ' | head -3"))
src = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8)      # 64 MiB block, reused
total = 8 * GB
blk = src.nbytes

def run(path, mode, nt, huge=False):
    if os.path.exists(path): os.unlink(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    nblk = total // blk
    mm = None
    if mode == "mmap":
        os.ftruncate(fd, total)
        mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        if huge:
            try: mm.madvise(mmap.MADV_HUGEPAGE)
            except Exception as e: print("madvise:", e)
        base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    nxt = [0]; lock = threading.Lock()
    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= nblk: return
            if mode == "pwrite": os.pwrite(fd, memoryview(src), i * blk)
            else: ctypes.memmove(base + i * blk, src.ctypes.data, blk)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work) for _ in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    if mm is not None:
        del base
        try: mm.close()
        except BufferError: pass
    os.close(fd); os.unlink(path)
    return total / dt / 1e9

for base in ("/dev/shm", "/tmp"):
    for mode in ("pwrite", "mmap"):
        for nt in (1, 4, 16):
            try:
                print(f"{base:9s} {mode:6s} {nt:2d} threads: {run(base + '/rcgpu_fsprobe.bin', mode, nt):6.2f} GB/s", flush=True)
            except Exception as e:
                print(base, mode, nt, "failed:", e)
    try: print(f"{base:9s} mmap+MADV_HUGEPAGE 16 threads: {run(base + '/rcgpu_fsprobe.bin', 'mmap', 16, True):6.2f} GB/s")
    except Exception as e: print("huge failed", e)
# tmpfs: pages allocated ahead by fallocate (one thread, under the inode lock, no copy), then 16 threads copy into the mapping
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def prealloc(path, nt, how="memcpy"):
    if os.path.exists(path): os.unlink(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    t0 = time.perf_counter(); os.posix_fallocate(fd, 0, total); t_fa = time.perf_counter() - t0
    mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    nblk = total // blk; nxt = [0]; lock = threading.Lock()
    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= nblk: return
            if how == "pwrite": os.pwrite(fd, memoryview(src), i * blk); continue
            if how == "populate": libc.madvise(base + i * blk, blk, 23)        # MADV_POPULATE_WRITE
            ctypes.memmove(base + i * blk, src.ctypes.data, blk)
    t0 = time.perf_counter(); th = [threading.Thread(target=work) for _ in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]; t_cp = time.perf_counter() - t0
    os.close(fd); os.unlink(path)
    return total / t_fa / 1e9, total / t_cp / 1e9
for base in ("/dev/shm", "/tmp"):
    try:
        for how, nt in (("memcpy", 16), ("memcpy", 4), ("populate", 16), ("pwrite", 16), ("pwrite", 1)):
            a, b = prealloc(base + "/rcgpu_fsprobe.bin", nt, how)
            print(f"{base:9s} fallocate 1 thread: {a:6.2f} GB/s, then {nt} threads {how}: {b:6.2f} GB/s", flush=True)
    except Exception as e:
        print(base, "prealloc failed:", e)
# 16 separate files, pwrite, one thread each: is the limit per inode?
def many(basedir, nt):
    fds = [os.open(f"{basedir}/rcgpu_fsprobe_{i}.bin", os.O_RDWR | os.O_CREAT, 0o644) for i in range(nt)]
    per = total // nt // blk
    def work(i):
        for k in range(per): os.pwrite(fds[i], memoryview(src), k * blk)
    t0 = time.perf_counter(); th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    for i, fd in enumerate(fds): os.close(fd); os.unlink(f"{basedir}/rcgpu_fsprobe_{i}.bin")
    return per * nt * blk / dt / 1e9
print(f"/dev/shm  16 files x pwrite: {many('/dev/shm', 16):6.2f} GB/s")
# reads: 32 files of 53 MB hard-linked 512 times, pread into a reused buffer
os.makedirs("/dev/shm/rcgpu_fsprobe_d", exist_ok=True)
n53 = 53084160 + 2048
for i in range(32):
    with open(f"/dev/shm/rcgpu_fsprobe_d/u{i}", "wb") as f: f.write(memoryview(src)[:n53])
names = []
for i in range(512):
    p = f"/dev/shm/rcgpu_fsprobe_d/f{i}"; os.link(f"/dev/shm/rcgpu_fsprobe_d/u{i % 32}", p); names.append(p)
for nt in (1, 4, 8, 16):
    nxt = [0]; lock = threading.Lock()
    def rd():
        buf = bytearray(n53)
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(names): return
            fd = os.open(names[i], os.O_RDONLY); os.preadv(fd, [buf], 0); os.close(fd)
    t0 = time.perf_counter(); th = [threading.Thread(target=rd) for _ in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print(f"tmpfs pread of 53 MB files, {nt:2d} threads: {len(names) * n53 / dt / 1e9:6.2f} GB/s = {len(names) / dt:6.1f} files/s")
import shutil; shutil.rmtree("/dev/shm/rcgpu_fsprobe_d")
