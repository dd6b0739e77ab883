"""tests -> oracle/_ref/ref_ffv1_decode (oracle/ref_ffv1_decode.cpp): the REAL reference's FFV1 decoder on a track's record and frames, no
Matroska file around them.  Present where oracle/Makefile.ref was run (this container; the binary travels to the GPU box)."""
import os
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_ffv1_decode")


def clean_env():
    """the environment for a driver: this process's without LD_PRELOAD (tools/asan_cpu.sh preloads the sanitizer runtime into the suite; the
    drivers are the reference's code, bound their own address space, and are not what is being sanitized)"""
    return {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}


def available() -> bool:
    return os.path.exists(EXE)


def decode(cases):
    """cases: [(flavor string, flags (1 VFlip | 2 Altern), width, height, record, [packets])] -> per case [(verdict, payload bytes)] per frame
    and the driver's line (the reference's first complaint at its end)."""
    blob = bytearray()
    for flavor, flags, w, h, rec, pks in cases:
        fl = flavor.encode()
        blob += struct.pack("<I", len(fl)) + fl + struct.pack("<IIII", flags, w, h, len(rec)) + rec + struct.pack("<I", len(pks))
        for p in pks:
            blob += struct.pack("<I", len(p)) + p
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "cases.bin"), os.path.join(t, "out.bin")
        open(a, "wb").write(blob)
        r = subprocess.run([EXE, a, b], capture_output=True, text=True, timeout=600, env=clean_env())
        assert r.returncode == 0, (r.returncode, r.stderr[-400:])
        out = open(b, "rb").read()
    lines, res, o = r.stdout.splitlines(), [], 0
    assert len(lines) == len(cases), r.stdout[-400:]
    for _, _, _, _, _, pks in cases:
        frames = []
        for _ in pks:
            v, n = struct.unpack_from("<II", out, o)
            frames.append((v, out[o + 8:o + 8 + n]))
            o += 8 + n
        res.append(frames)
    return res, lines


FLAC_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_flac_decode")


def flac_available() -> bool:
    return os.path.exists(FLAC_EXE)


def flac_decode(cases):
    """cases: [(bits of the WAV, CodecPrivate, [FLAC frames])] -> [(complaints, PCM bytes)]: the reference's own FLAC wrapper and the libFLAC
    it ships (oracle/ref_flac_decode.cpp)."""
    blob = bytearray()
    for bits, cp, frames in cases:
        blob += struct.pack("<II", bits, len(cp)) + cp + struct.pack("<I", len(frames))
        for fr in frames:
            blob += struct.pack("<I", len(fr)) + fr
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "cases.bin"), os.path.join(t, "out.bin")
        open(a, "wb").write(blob)
        r = subprocess.run([FLAC_EXE, a, b], capture_output=True, text=True, timeout=600, env=clean_env())
        assert r.returncode == 0, (r.returncode, r.stderr[-400:])
        out = open(b, "rb").read()
    res, o = [], 0
    for _ in cases:
        v, n = struct.unpack_from("<II", out, o)
        res.append((v, out[o + 8:o + 8 + n]))
        o += 8 + n
    return res


def run_reference(cmd, cwd, timeout=60, attempts=3):
    """Start the real reference binary (or the linked one).  Retried on a timeout, for one known cause in unpatched reference code: the lost
    wake-up of its third-party thread pool at shutdown (Lib/ThirdParty/thread-pool/include/ThreadPool.h:27-33,66-68; tests/test_gpu_e2e.py::run
    has the whole story) -- about one run in a hundred on a loaded box never returns.  Every step the tests make is idempotent."""
    import sys
    for attempt in range(attempts):
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=timeout, env=clean_env())
        except subprocess.TimeoutExpired:
            sys.stderr.write("reference step timed out after %d s (attempt %d): %s\n" % (timeout, attempt + 1, " ".join(cmd[:5])))
            if attempt + 1 == attempts:
                raise
