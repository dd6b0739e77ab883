#!/usr/bin/env python3
"""Vectors that pin the header probes against the REAL reference's parsers (tests/golden/probe_cases.txt).

  make -f oracle/Makefile.ref probe && python tests/golden/make_probe_golden.py

Every case is the bytes of a small file: DPX of every flavor, TIFF, EXR, WAV as rawcooked_amd/synth.py writes them, and seeded mutations of
their headers (bytes overwritten, bits flipped, files cut short).  oracle/_ref/ref_probe shows each to the reference's wav, dpx, tiff and exr
parsers in the order CLI/Main.cpp tries them and prints which one recognised it, whether it supports it, the flavor string and slice_x *
slice_y; tests/test_host.py::test_probes_agree_with_the_reference_s_parsers holds rcgpu_*_probe to those lines: what the reference would hand
to its encoder, the shim must take, as the same flavor, with the same slice count.  Deterministic: only the reference's lines are kept (the
first one is the sha256 of all the cases' bytes); the test makes the same 3601 files again with cases() below.
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rawcooked_amd import synth  # noqa: E402


def cases():
    rng = np.random.default_rng(20261002)
    s = rng.integers(-1000, 1000, size=(300, 2)).astype(np.int32)
    seeds = []
    P = synth
    # every DPX layout the library knows (the first eight were the fixture's first edition: their order is kept), widths its blocks divide
    for pf, (w, h) in [(P.PIX_RGB16_BE, (32, 16)), (P.PIX_RGB10_FILLEDA_BE, (33, 7)), (P.PIX_RGB12_PACKED_BE, (32, 9)), (P.PIX_RGBA16_LE, (24, 10)),
                       (P.PIX_Y16_BE, (40, 12)), (P.PIX_RGB8, (48, 8)), (P.PIX_RGB10_FILLEDA_LE, (31, 9)), (P.PIX_RGBA8, (20, 6))]:
        bits, nc, _, _ = synth.PIX_INFO[pf]
        seeds.append(synth.dpx_file(synth.components(w, h, nc, bits, "film", seed=1), pf))
    seeds.append(synth.tiff_file(synth.components(32, 16, 3, 16, "film", seed=1), synth.PIX_RGB16_LE, trailer=b"xx"))
    seeds.append(synth.tiff_file(synth.components(30, 11, 3, 8, "film", seed=1), synth.PIX_RGB8))
    seeds.append(synth.exr_file(synth.components(32, 16, 3, 16, "film", seed=1)))
    seeds.append(synth.wav_file(s, 16))
    seeds.append(synth.wav_file(np.tile(s, (1, 3)), 24, extensible=True))
    more = []
    for pf, (w, h) in [(P.PIX_RGB12_FILLEDA_BE, (32, 8)), (P.PIX_RGB12_FILLEDA_LE, (28, 8)), (P.PIX_RGB16_LE, (32, 8)), (P.PIX_RGBA16_BE, (24, 8)), (P.PIX_Y8, (64, 8)), (P.PIX_Y16_LE, (40, 8)),
                       (P.PIX_RGBA10_FILLEDA_BE, (48, 8)), (P.PIX_RGBA10_FILLEDA_LE, (48, 8)), (P.PIX_RGBA12_PACKED_BE, (48, 8)), (P.PIX_RGBA12_FILLEDA_BE, (24, 8)), (P.PIX_RGBA12_FILLEDA_LE, (24, 8)),
                       (P.PIX_Y10_FILLEDA_BE, (48, 8)), (P.PIX_Y10_FILLEDB_BE, (48, 8)), (P.PIX_Y12_PACKED_BE, (96, 8))]:
        bits, nc, _, _ = synth.PIX_INFO[pf]
        more.append(synth.dpx_file(synth.components(w, h, nc, bits, "film", seed=1), pf))
    for pf in (P.PIX_RGB16_BE, P.PIX_RGBA8, P.PIX_RGBA16_LE, P.PIX_Y8, P.PIX_Y16_LE, P.PIX_Y16_BE):
        bits, nc, _, _ = synth.PIX_INFO[pf]
        more.append(synth.tiff_file(synth.components(32, 8, nc, bits, "film", seed=1), pf))
    more.append(synth.wav_file(s[:, :1], 8))
    more.append(synth.wav_file(np.tile(s, (1, 4)), 16, 96000))
    more.append(synth.wav_file(s, 32))                                    # what only `-c:a copy` can carry (CLI/Main.cpp:300-317): 32-bit integer
    more.append(synth.wav_file(s, 32, float32=True))                      # ... and 32-bit float
    more.append(synth.wav_file(np.tile(s, (1, 2)), 32, 44100, extensible=True, float32=True))
    more.append(synth.wav_file(s[:, :1], 24, 44100, trailer_chunk=b"LIST" + (4).to_bytes(4, "little") + b"abcd"))
    out = list(seeds)
    for d in seeds:
        for k in range(150):
            b = bytearray(d)
            if k % 5 == 4:
                b = b[:int(rng.integers(0, len(b)))]
            for _ in range(int(rng.integers(1, 4))):
                if len(b):
                    at = int(rng.integers(0, min(len(b), 2100)))
                    b[at] = int(rng.integers(0, 256)) if rng.integers(0, 2) else b[at] ^ (1 << int(rng.integers(0, 8)))
            out.append(bytes(b))
    # the fields the first differential run found the probes stricter on than the reference: DPX orientation (only 2 means anything to it,
    # DPX.cpp:411-412), the high byte of the DPX packing field (an 8-bit enum there, DPX.cpp:134,348), the VALUE of the TIFF Compression tag
    # (only its presence is tested, TIFF.cpp:465,557-558)
    for d in seeds[:8]:
        be = d[:4] == b"SDPX"
        for o in (1, 3, 7, 80, 0xFFFF):
            b = bytearray(d); b[768:770] = o.to_bytes(2, "big" if be else "little"); out.append(bytes(b))
        b = bytearray(d); b[804 if be else 805] ^= 0x02; out.append(bytes(b))
    for d in seeds[11:13]:                                    # WAV: AvgBytesPerSec x 8 is a 32-bit product there (WAV.cpp:476): the top bits wrap away
        b = bytearray(d); b[31] ^= 0x80; out.append(bytes(b))
        b = bytearray(d); b[31] ^= 0x20; out.append(bytes(b))
    # second edition: the other fourteen DPX layouts, six more TIFF ones, two more WAVs -- as they are and with 60 mutations each
    out += more
    for d in more:
        for k in range(60):
            b = bytearray(d)
            if k % 6 == 5:
                b = b[:int(rng.integers(0, len(b)))]
            for _ in range(int(rng.integers(1, 4))):
                if len(b):
                    at = int(rng.integers(0, min(len(b), 2100)))
                    b[at] = int(rng.integers(0, 256)) if rng.integers(0, 2) else b[at] ^ (1 << int(rng.integers(0, 8)))
            out.append(bytes(b))
    return out


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    cs = cases()
    blob = b"".join(struct.pack("<I", len(c)) + c for c in cs)
    with tempfile.TemporaryDirectory() as t:
        open(os.path.join(t, "cases.bin"), "wb").write(blob)
        out = subprocess.run([exe, os.path.join(t, "scratch.rev"), os.path.join(t, "cases.bin")], capture_output=True, text=True, check=True, cwd=t).stdout
    lines = out.splitlines()
    assert len(lines) == len(cs), (len(lines), len(cs))
    open(os.path.join(HERE, "probe_cases.txt"), "w").write("sha256 " + hashlib.sha256(blob).hexdigest() + "\n" + out)
    print(f"{len(cs)} cases ({len(blob)} bytes): the reference supports {sum('supported=1' in x for x in lines)}")


if __name__ == "__main__":
    main()
