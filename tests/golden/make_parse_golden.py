#!/usr/bin/env python3
"""Vectors that pin the FFV1 header readers against the REAL reference's (tests/golden/parse_cases.bin + parse_cases.txt).

  make -f oracle/Makefile.ref ffv1_parse && python tests/golden/make_parse_golden.py

Every case is (configuration record, first bytes of the first frame) -- or (nothing, a version 0 / 1 key frame).  oracle/_ref/ref_ffv1_parse
feeds each to the reference's own ffv1_frame::OutOfBand / parameters::Parse / slice::SliceHeader (FFV1_Frame.cpp:105-131, FFV1_Parameters.cpp:
23-183, FFV1_Slice.cpp:113-177) and prints what they made of it; tests/test_host.py::test_stream_parse_agrees_with_the_reference_s_reader holds
rcgpu_ffv1_stream_parse to those lines.  The cases: every golden record and the version 0 / 1 headers as they are; records WRITTEN field by
field (tests/range_writer.py) with values on and beyond every limit the reference tests; and seeded mutations of all of them, re-sealed with a
valid CRC.  Deterministic: the same script gives the same files.
"""
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import range_writer as rw  # noqa: E402

MAX_REC = 700          # larger golden records (coded initial states: 6-12 KB) are cut to their first bytes and re-sealed: hostile, and small


def crafted():
    """(record, first bytes of the frame) pairs written field by field: one value at, below and beyond each limit parameters::Parse and
    slice::SliceHeader test."""
    R, H = rw.record, rw.first_slice
    runs128 = [0] * 128                                       # 128 runs of one: 128 levels, scale x 255
    d1, dr = [1] * 255, [int(x) for x in np.random.default_rng(3).integers(-2, 3, size=255)]
    P = []

    def add(rec, idx=(0, 0), **kw):
        P.append((rec, H(idx, **kw)))
    for rec in (R(), R(version=0), R(version=1), R(version=2), R(version=4), R(version=7), R(micro=0), R(micro=3), R(micro=4), R(micro=5), R(micro=9), R(coder=0), R(coder=3),
                R(coder=2, deltas=[-200] + [0] * 254), R(coder=2, deltas=[0] * 254 + [20]), R(colorspace=2), R(bits=0), R(bits=8), R(bits=10), R(bits=64), R(bits=65), R(bits=1 << 20),
                R(colorspace=0, chroma=0), R(colorspace=0, chroma=1, log2h=1, log2v=1), R(h1=0xFFFFFFFF, v1=0), R(h1=1 << 20, v1=1 << 20), R(sets=[]), R(sets=[[[127]] * 5] * 9),
                R(sets=[[[126], [127], [127], [127], [127]]]), R(sets=[[[128], [127], [127], [127], [127]]]), R(sets=[[[0, 126], [127], [127], [127], [127]]]),
                R(sets=[[[0, 127], [127], [127], [127], [127]]]), R(sets=[[[0, 0xFFFFFFFF], [127], [127], [127], [127]]]), R(sets=[[[0, 0xFFFFFF7F], [127], [127], [127], [127]]]),
                R(sets=[[[1 << 32], [127], [127], [127], [127]]]), R(sets=[[[63, 63]] * 5]), R(sets=[[[31, 31, 31, 31]] * 5]),
                R(sets=[[runs128, [127], [127], [127], [127]]]), R(sets=[[runs128, runs128, [127], [127], [127]]]),          # 255; 255 x 255 = 65025 > 32768
                R(sets=[[runs128, [42, 42, 41], [127], [127], [127]]]),                                                    # 255 x 5 = 1275
                R(sets=[[[0] * 90 + [37], [0] * 60 + [67], [127], [127], [127]]]),                                         # 181 x 121 = 21901
                R(sets=[[[0] * 90 + [37], [0] * 90 + [37], [127], [127], [127]]]),                                         # 181 x 181 = 32761: just fits
                R(sets=[[[0] * 90 + [37], [0] * 91 + [36], [127], [127], [127]]]),                                         # 181 x 183 = 33123: does not
                R(ec=0), R(ec=1), R(ec=2), R(intra=0), R(intra=2), R(micro=0, intra=1),
                R(sets=[[[63, 63]] * 5], coded=[[128] * (122 * 32)]), R(sets=[[[63, 63]] * 5], coded=[[1, 255, -1, 300, -300] * 10]),
                R(tail=b"\x00" * 40), R(tail=bytes(range(200)))):
        add(rec)
    add(R(coder=2, deltas=d1), one_state=rw.one_state_of(d1)); add(R(coder=2, deltas=dr), one_state=rw.one_state_of(dr)); add(R(coder=2, deltas=dr))      # (the last: header under the wrong table)
    add(R(colorspace=1, alpha=1), (0, 0, 0)); add(R(colorspace=0, chroma=0, alpha=1), (0, 0, 0)); add(R(colorspace=1, alpha=1), (0, 0))
    for h1, v1 in ((0, 0), (31, 17), (17, 31), (2, 5), (5, 2), (0xFFFE, 0), (0, 0xFFFE)):
        add(R(h1=h1, v1=v1))
        add(R(h1=h1, v1=v1), xywh=(h1, 0, 0, 0)); add(R(h1=h1, v1=v1), xywh=(0, v1, 0, 0)); add(R(h1=h1, v1=v1), xywh=(0, 0, h1, v1)); add(R(h1=h1, v1=v1), xywh=(1, 1, h1, v1))
        add(R(h1=h1, v1=v1), xywh=(h1 + 1, 0, 0, 0)); add(R(h1=h1, v1=v1), xywh=(0, v1 + 1, 0, 0)); add(R(h1=h1, v1=v1), xywh=(0, 0, 0xFFFFFFFF, 0)); add(R(h1=h1, v1=v1), xywh=(1, 0, 0xFFFFFFFF, 0))
    eight = [[[127]] * 5] * 8
    add(R(sets=eight), (7, 3)); add(R(sets=eight), (8, 0)); add(R(sets=eight), (0, 8)); add(R(sets=eight), (0xFFFFFFFF, 0)); add(R(sets=eight), (0, 0), key=0)
    add(R(sets=eight), (3, 4), tail=b""); add(R(sets=[[[63, 63]] * 5] * 2, coded=[None, [int(x) for x in np.random.default_rng(5).integers(1, 255, size=122 * 32)]]), (1, 0))
    for k in range(1, 13):                                                                                                 # records that stop after k fields
        add(R(upto=k)); add(R(upto=k, tail=b"\xff" * 30)); add(R(upto=k, tail=b"\x00" * 30))
    return P


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_ffv1_parse")
    V = json.load(open(os.path.join(HERE, "vectors.json")))
    rng = np.random.default_rng(20261001)
    seeds = []                       # (record, packet head)
    for v in V["ffv1_ext"] + V["ffv1"]:
        rec = open(os.path.join(HERE, v["config_record_file"]), "rb").read() if "config_record_file" in v else bytes.fromhex(v.get("config_record", ""))
        pk = open(os.path.join(HERE, v["frames"][0]["packet"]), "rb").read()
        if len(rec) > MAX_REC:
            rec = rw.sealed(rec[:MAX_REC - 4])
        seeds.append((rec, pk[:48] if rec else pk[:MAX_REC]))
    seeds += crafted()
    cases = list(seeds)
    for rec, pk in seeds:
        for k in range(14):
            r, q = bytearray(rec), bytearray(pk)
            tgt = r if (rec and k % 4) else q
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, max(1, len(tgt) - (4 if tgt is r else 0))))
                tgt[at] = int(rng.integers(0, 256)) if rng.integers(0, 2) else tgt[at] ^ (1 << int(rng.integers(0, 8)))
            if rec and k % 7 == 3:
                r = r[:int(rng.integers(5, len(r) + 1))]
            if rec and k % 9 != 8:                               # (one in nine keeps its broken CRC)
                r = bytearray(rw.sealed(bytes(r[:-4])))
            cases.append((bytes(r), bytes(q)))
    blob = b"".join(struct.pack("<II", len(r), len(p)) + r + p for r, p in cases)
    open(os.path.join(HERE, "parse_cases.bin"), "wb").write(blob)
    out = subprocess.run([exe, os.path.join(HERE, "parse_cases.bin")], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    assert len(lines) == len(cases), (len(lines), len(cases))
    open(os.path.join(HERE, "parse_cases.txt"), "w").write(out)
    ok = sum(line.startswith("OK") for line in lines)
    print(f"{len(cases)} cases ({len(blob)} bytes): the reference read {ok} and refused {len(cases) - ok}")


if __name__ == "__main__":
    main()
