"""Pins the predictor words of the golden FLAC streams: per frame the channel assignment and per subframe (type, order, precision, shift,
quantised coefficients, Rice partition order), read out of tests/golden/flac_*.frames.bin with the specification-level decoder
tests/flac_validator.py -> tests/golden/flac_coefficients.json.

Why: Levinson-Durbin and the coefficient quantiser run in IEEE double with a fixed operation order on the device (-ffp-contract=off) and
must round like the scalar oracle (reference maths: Lib/ThirdParty/flac/src/libFLAC/lpc.c:122-259).  A compiler update that moves one
rounding would change a coefficient word; with this file the tests name the block, channel and coefficient that moved instead of
reporting a frame of a different size.  Run in the build container:  python tests/golden/make_flac_coefficients.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import flac_validator as fv  # noqa: E402

vec = json.load(open(os.path.join(HERE, "vectors.json")))
out = {}
for f in vec["flac"]:
    frames = open(os.path.join(HERE, f["frames"]), "rb").read()
    pcm, infos = fv.parse_stream(frames, f["channels"], f["bits"], f["rate"])
    assert pcm == open(os.path.join(HERE, f["pcm"]), "rb").read(), f["name"]
    out[f["name"]] = fv.predictor_words(infos)
with open(os.path.join(HERE, "flac_coefficients.json"), "w") as fh:
    json.dump(out, fh, indent=0, separators=(",", ":"))
print("wrote flac_coefficients.json:", {k: len(v) for k, v in out.items()})
