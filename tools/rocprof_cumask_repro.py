"""rocprofv3 --kernel-trace and this library's CU-masked streams: what it takes to lose the trace at exit (SIGSEGV under __cxa_finalize).
The C++ program next door (rocprof_cumask_repro.hip: a masked stream, kernels on it, alive or not at exit) survives every rocprofv3 mode.
  python tools/rocprof_cumask_repro.py <variant>
    lib        ctypes only: librcgpu's rcgpu_md5_host_batch (its per-device hash stream is CU-masked), nothing else
    lib+free   the same, the stream given back before exit (rcgpu_release_device_streams)
    torch+lib  `import torch` first, then the same
    torch      torch alone: one kernel, no masked stream
    torch+mask torch + a CU-masked stream made through ctypes (hipExtStreamCreateWithCUMask), a memset on it
`bash tools/round.sh cumaskpy` runs each under rocprofv3 --kernel-trace --stats and records the exit status."""
import ctypes
import os
import sys

v = sys.argv[1] if len(sys.argv) > 1 else "lib"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if v.startswith("torch"):
    import torch
    x = torch.ones(1 << 20, device="cuda"); x = (x * 2).sum().item()
if "lib" in v:
    from rawcooked_amd import api
    import hashlib
    buf = bytes(range(256)) * 4096
    assert api.md5_host_batch([buf, buf[:1000]]) == [hashlib.md5(buf).digest(), hashlib.md5(buf[:1000]).digest()]
    if "free" in v:
        api.lib().rcgpu_release_device_streams()
if "mask" in v:
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * 8)(0xFF, 0, 0, 0, 0, 0, 0, 0)
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, mask) == 0
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), 1 << 20) == 0
    assert hip.hipMemsetAsync(p, 1, 1 << 20, st) == 0 and hip.hipStreamSynchronize(st) == 0
print("ok:", v)
