#!/usr/bin/env python3
"""Generates tests/golden/padding_vectors.json with the REAL reference parser: DPX files of every layout that can carry padding bits
(DPX.cpp:184-207 rows with FilledA / FilledB / Packed), clean and with bits poked into samples, filler bits, line padding and the last
word, go through oracle/_ref/ref_padding_probe (= dpx::ParseBuffer with --check-padding, built by `make -f oracle/Makefile.ref
padding_probe`), which prints the parser's own In_FirstNonZero (DPX.cpp:501-608).  The vectors store only the recipe (layout, size,
seed, pokes) and the reference's answer; the tests rebuild the payloads with rawcooked_amd/synth.py.

Run here (needs /root/reference):   python tests/golden/make_padding_golden.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rawcooked_amd import synth   # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_padding_probe")
LAYOUTS = [synth.PIX_RGB10_FILLEDA_BE, synth.PIX_RGB10_FILLEDA_LE, synth.PIX_RGB12_FILLEDA_BE, synth.PIX_RGB12_FILLEDA_LE, synth.PIX_RGB12_PACKED_BE,
           synth.PIX_RGBA10_FILLEDA_BE, synth.PIX_RGBA10_FILLEDA_LE, synth.PIX_RGBA12_PACKED_BE, synth.PIX_RGBA12_FILLEDA_BE, synth.PIX_RGBA12_FILLEDA_LE,
           synth.PIX_Y10_FILLEDA_BE, synth.PIX_Y10_FILLEDB_BE, synth.PIX_Y12_PACKED_BE, synth.PIX_RGB16_BE, synth.PIX_RGB8]


def payload_for(v):
    bits, nc, _, _ = synth.PIX_INFO[v["pixfmt"]]
    pl, _ = synth.pack_payload(synth.components(v["width"], v["height"], nc, bits, "noise", seed=v["seed"]), v["pixfmt"], True, v["flags"])
    b = bytearray(len(pl)) if v["zero"] else bytearray(pl)
    for at, x in v["pokes"]:
        b[at] ^= x
    return bytes(b)


def main():
    if not os.path.exists(PROBE):
        sys.exit("build it first: make -f oracle/Makefile.ref padding_probe")
    rng = np.random.default_rng(2026)
    vectors = []
    for pixfmt in LAYOUTS:
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        for (w, h) in ((37, 9), (64, 5), (50, 7), (4, 3)):
            if pixfmt in (synth.PIX_RGB12_FILLEDA_BE, synth.PIX_RGB12_FILLEDA_LE) and w % 2:
                continue      # the reference sizes the payload of odd-width RGB 12-bit FilledA without the line padding its own writer adds
                              # (DPX.cpp:470-474 vs RawFrame.cpp:109): not a geometry it round-trips, not a vector
            alterns = (0, synth.FLAG_ALTERN) if pixfmt in (synth.PIX_Y10_FILLEDA_BE, synth.PIX_Y10_FILLEDB_BE) else (0,)
            for flags in alterns:
                base = dict(pixfmt=pixfmt, width=w, height=h, flags=flags, seed=w * 131 + h)
                n = len(payload_for(dict(base, zero=False, pokes=[])))
                vectors.append(dict(base, zero=False, pokes=[]))
                vectors.append(dict(base, zero=True, pokes=[]))
                for k in range(10):
                    pokes = []
                    for _ in range(int(rng.integers(1, 4))):
                        at = int(rng.integers(0, n)) if k % 3 else n - 1 - int(rng.integers(0, min(n, 8)))
                        pokes.append([at, 1 << int(rng.integers(0, 8))])
                    vectors.append(dict(base, zero=bool(k & 1), pokes=pokes))
    with tempfile.TemporaryDirectory() as work:
        names = []
        for i, v in enumerate(vectors):
            p = os.path.join(work, "v%05d.dpx" % i)
            with open(p, "wb") as f:
                f.write(synth.dpx_file(None, v["pixfmt"], payload=payload_for(v), size=(v["width"], v["height"]), flags=v["flags"]))
            names.append(p)
        out = []
        for i in range(0, len(names), 200):
            r = subprocess.run([PROBE, os.path.join(work, "scratch.rev")] + names[i:i + 200], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            out += [ln for ln in r.stdout.splitlines() if ln.startswith(work)]
    assert len(out) == len(vectors), (len(out), len(vectors))
    kept = []
    for v, ln in zip(vectors, out):
        f = dict(x.split("=", 1) for x in ln.split()[1:])
        if f["supported"] != "1":
            continue                                          # a geometry the reference refuses for this flavor: not a vector
        v["flavor"] = f["flavor"]
        v["in_size"] = int(f["in_size"])
        v["first_nonzero"] = int(f["first_nonzero"])        # -1: every padding bit is zero, no In block
        kept.append(v)
    with open(os.path.join(HERE, "padding_vectors.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_padding_golden.py with oracle/_ref/ref_padding_probe (the reference's dpx::ParseBuffer, DPX.cpp:501-608)",
                   "vectors": kept}, f, separators=(",", ":"))
    print(len(kept), "vectors,", sum(1 for v in kept if v["first_nonzero"] >= 0), "with non-zero padding bits,", len(vectors) - len(kept), "refused")


if __name__ == "__main__":
    main()
