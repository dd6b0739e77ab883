"""Where do k_resolve's cycles go?  Needs the timing build with the probes: `make -C rawcooked_amd/csrc timing PROF=1`,
then RCGPU_LIB=rawcooked_amd/librcgpu_timing.so (s_memtime around s_memtime around the phases of a chunk,
summed over all wavefronts).  Usage on the GPU box:  python tools/prof_resolve.py [bench.py arguments]"""
import ctypes as C
import io
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rawcooked_amd import api  # noqa: E402

sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--legs", "", "--no-verify"] + sys.argv[1:]
out = (C.c_ulonglong * 16)()
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
assert api.lib().rcgpu_debug_prof(out) == 0       # loaded by bench after torch's HIP runtime
names = ["scan/decode symbol", "collision analysis", "vmcnt(0) wait", "install + prefetch issue", "coded bits", "rounds: rest (LDS drain, loop)", "state write-back", "flush", None, "  binarise: load + phase 1", "  binarise: chains", "  binarise: phase 2"]
chunks = out[8]
tot = sum(out[i] for i in range(12) if i != 8)
print("chunks %d, cycles per chunk %.0f" % (chunks, tot / max(1, chunks)))
for i, n in enumerate(names):
    if n is None: continue
    print("  %-26s %8.0f cycles per chunk  %5.1f %%" % (n, out[i] / max(1, chunks), 100.0 * out[i] / max(1, tot)))
