#!/usr/bin/env python3
"""Do many pinned H2D copies on one stream and many D2H copies on another overlap, or do they queue behind each other?
(The pipeline issues a batch of downloads, then the next batch's uploads.)"""
import ctypes, time
import torch

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
N, SZ_IN, SZ_OUT = 256, 53084160, 48847592
d_in = torch.empty(N * SZ_IN, dtype=torch.uint8, device=dev)
d_out = torch.empty(N * SZ_OUT, dtype=torch.uint8, device=dev)

def pinned(n, flags):
    p = ctypes.c_void_p()
    assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), ctypes.c_uint(flags)) == 0
    return p.value

for name, flags in (("hipHostMallocDefault", 0), ("hipHostMallocPortable", 1), ("Portable|NonCoherent", 1 | 0x80000000)):
    try:
        h_in = [pinned(SZ_IN, flags) for _ in range(32)]
        h_out = [pinned(SZ_OUT, flags) for _ in range(32)]
    except AssertionError:
        print(name, "allocation failed"); continue
    s_up, s_dn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    H2D, D2H = 1, 2
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    def up():
        for i in range(N):
            assert hip.hipMemcpyAsync(d_in.data_ptr() + i * SZ_IN, h_in[i % 32], SZ_IN, H2D, s_up.cuda_stream) == 0
    def dn():
        for i in range(N):
            assert hip.hipMemcpyAsync(h_out[i % 32], d_out.data_ptr() + i * SZ_OUT, SZ_OUT, D2H, s_dn.cuda_stream) == 0
    for label, first, second in (("uploads alone", up, None), ("downloads alone", dn, None), ("downloads issued first, then uploads", dn, up), ("uploads first, then downloads", up, dn)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first(); t_issue1 = time.perf_counter() - t0
        if second: second()
        t_issue = time.perf_counter() - t0
        s_up.synchronize(); t_up = time.perf_counter() - t0
        s_dn.synchronize(); t_dn = time.perf_counter() - t0
        print(f"{name:24s} {label:40s} issue {t_issue1:.3f}/{t_issue:.3f} s, upload stream idle after {t_up:.3f} s, download stream after {t_dn:.3f} s "
              f"(alone they would take {N * SZ_IN / 55e9:.3f} / {N * SZ_OUT / 55e9:.3f} s)", flush=True)

# ---- the pipeline's pattern: an event behind every group of copies, uploads trickling in as "slots" (64) complete
print("--- with events ---")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventQuery.argtypes = [ctypes.c_void_p]
def mkev():
    e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), 0x2) == 0; return e      # hipEventDisableTiming
h_in = [pinned(SZ_IN, 1) for _ in range(64)]
h_out = [pinned(SZ_OUT, 1) for _ in range(64)]
for label, ev_dn, ev_up, busy in (("events on both streams", True, True, False), ("events on uploads only", False, True, False), ("events on downloads only", True, False, False),
                                  ("events on both + a kernel-busy device", True, True, True)):
    s_up, s_dn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    if busy:
        sb = torch.cuda.Stream(device=dev)
        a = torch.randn(8192, 8192, device=dev)
        with torch.cuda.stream(sb):
            for _ in range(60): a = a @ a * 1e-4
    torch.cuda.synchronize() if not busy else None
    t0 = time.perf_counter()
    evs_dn = []
    for i in range(N):
        assert hip.hipMemcpyAsync(h_out[i % 64], d_out.data_ptr() + i * SZ_OUT, SZ_OUT, D2H, s_dn.cuda_stream) == 0
        if ev_dn and i % 16 == 15:
            e = mkev(); hip.hipEventRecord(e, s_dn.cuda_stream); evs_dn.append(e)
    first_done = None
    evs_up = []
    for i in range(N):
        assert hip.hipMemcpyAsync(d_in.data_ptr() + i * SZ_IN, h_in[i % 64], SZ_IN, H2D, s_up.cuda_stream) == 0
        if ev_up and i % 8 == 7:
            e = mkev(); hip.hipEventRecord(e, s_up.cuda_stream); evs_up.append(e)
    t_issue = time.perf_counter() - t0
    if evs_up:
        while hip.hipEventQuery(evs_up[0]) != 0: time.sleep(0.0005)
        first_done = time.perf_counter() - t0
    s_up.synchronize(); t_up = time.perf_counter() - t0
    s_dn.synchronize(); t_dn = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{label:44s} issued in {t_issue:.3f} s, first upload group done after {first_done if first_done is None else round(first_done, 3)} s, uploads done {t_up:.3f} s, downloads done {t_dn:.3f} s", flush=True)

# ---- closer to the pipeline: raw HIP streams created after several others, odd packet sizes, a stream-wait in front of the downloads
print("--- raw streams, odd sizes, stream wait ---")
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
def mkstream():
    st = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0; return st       # hipStreamNonBlocking
others = [mkstream() for _ in range(2)]
cin, cout = mkstream(), mkstream()
for label, odd, wait, nstreams_busy in (("aligned sizes", False, False, 0), ("odd sizes", True, False, 0), ("odd sizes + stream wait", True, True, 0),
                                        ("odd + wait + two busy compute streams", True, True, 2)):
    busy = []
    if nstreams_busy:
        x = [torch.randn(64 << 20, device=dev) for _ in range(nstreams_busy)]
        for k in range(nstreams_busy):
            sb = torch.cuda.Stream(device=dev); busy.append(sb)
            with torch.cuda.stream(sb):
                for _ in range(400): x[k] = torch.cumsum(x[k], 0) * 1e-9
    e0 = mkev(); hip.hipEventRecord(e0, others[0])
    t0 = time.perf_counter()
    if wait: hip.hipStreamWaitEvent(cout, e0, 0)
    for i in range(N):
        sz = SZ_OUT - (i * 7919 % 1000) if odd else SZ_OUT
        assert hip.hipMemcpyAsync(h_out[i % 64], d_out.data_ptr() + i * SZ_OUT, sz, D2H, cout) == 0
        if i % 16 == 15: e = mkev(); hip.hipEventRecord(e, cout)
    evs_up = []
    for i in range(N):
        assert hip.hipMemcpyAsync(d_in.data_ptr() + i * SZ_IN, h_in[i % 64], SZ_IN, H2D, cin) == 0
        if i % 8 == 7: e = mkev(); hip.hipEventRecord(e, cin); evs_up.append(e)
    while hip.hipEventQuery(evs_up[0]) != 0: time.sleep(0.0005)
    first_done = time.perf_counter() - t0
    hip.hipStreamSynchronize(cin); t_up = time.perf_counter() - t0
    hip.hipStreamSynchronize(cout); t_dn = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{label:44s} first upload group done after {first_done:.3f} s, uploads done {t_up:.3f} s, downloads done {t_dn:.3f} s", flush=True)

# ---- with CPU threads working on the same pinned buffers, as the pipeline's readers and writers do
print("--- CPU threads copy into the upload buffers / out of the download buffers meanwhile ---")
import threading, numpy as np
src = np.random.default_rng(0).integers(0, 255, SZ_IN, dtype=np.uint8)
sink = [np.empty(SZ_OUT, dtype=np.uint8) for _ in range(8)]
for label, nr, nw in (("no CPU threads", 0, 0), ("8 readers", 8, 0), ("8 writers", 0, 8), ("8 readers + 8 writers", 8, 8)):
    stop = threading.Event()
    def rd(k):
        i = k
        while not stop.is_set():
            ctypes.memmove(h_in[i % 64], src.ctypes.data, SZ_IN); i += 8
    def wr(k):
        i = k
        while not stop.is_set():
            ctypes.memmove(sink[k].ctypes.data, h_out[i % 64], SZ_OUT); i += 8
    ths = [threading.Thread(target=rd, args=(k,)) for k in range(nr)] + [threading.Thread(target=wr, args=(k,)) for k in range(nw)]
    [t.start() for t in ths]
    time.sleep(0.2)
    t0 = time.perf_counter()
    for i in range(N):
        assert hip.hipMemcpyAsync(h_out[i % 64], d_out.data_ptr() + i * SZ_OUT, SZ_OUT, D2H, cout) == 0
        if i % 16 == 15: e = mkev(); hip.hipEventRecord(e, cout)
    evs_up = []
    for i in range(N):
        assert hip.hipMemcpyAsync(d_in.data_ptr() + i * SZ_IN, h_in[i % 64], SZ_IN, H2D, cin) == 0
        if i % 8 == 7: e = mkev(); hip.hipEventRecord(e, cin); evs_up.append(e)
    while hip.hipEventQuery(evs_up[0]) != 0: time.sleep(0.0005)
    first_done = time.perf_counter() - t0
    hip.hipStreamSynchronize(cin); t_up = time.perf_counter() - t0
    hip.hipStreamSynchronize(cout); t_dn = time.perf_counter() - t0
    stop.set(); [t.join() for t in ths]
    print(f"{label:44s} first upload group done after {first_done:.3f} s, uploads done {t_up:.3f} s, downloads done {t_dn:.3f} s", flush=True)
