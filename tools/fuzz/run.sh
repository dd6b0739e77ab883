#!/bin/bash
# Mutation fuzzing of everything that parses bytes it did not write, on the CPU under ASan + UBSan: the DPX / TIFF / EXR / WAV
# probes, the FFV1 configuration-record / stream reader (mutated records are re-sealed with a valid CRC), and the job front end -- the argv grammar and
# the ffconcat lists the reference writes (CLI/Output.cpp:81-332) -- in plan-only mode.  Usage: bash tools/fuzz/run.sh [iterations per seed] [argv iterations]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d); trap 'rm -rf $T' EXIT
N=${1:-200000}
NA=${2:-1000000}
mkdir -p $T/seeds $T/rec $T/job/seq
cd $R
python - $T <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_binding as ob
from rawcooked_amd import synth
t = sys.argv[1]
comp = synth.components(32, 16, 3, 16, "film", seed=1)
w = lambda n, b: open(os.path.join(t, n), "wb").write(b)
w("seeds/a.dpx", synth.dpx_file(comp, synth.PIX_RGB16_BE))
w("seeds/b.dpx", synth.dpx_file(synth.components(33, 7, 3, 10, "film", seed=2), synth.PIX_RGB10_FILLEDA_BE))
w("seeds/c.dpx", synth.dpx_file(synth.components(30, 9, 3, 12, "film", seed=2), synth.PIX_RGB12_PACKED_BE))
w("seeds/a.tif", synth.tiff_file(comp, synth.PIX_RGB16_LE, trailer=b"xx"))
w("seeds/a.exr", synth.exr_file(comp))
s = np.random.default_rng(1).integers(-1000, 1000, size=(500, 2)).astype(np.int32)
w("seeds/a.wav", synth.wav_file(s, 16))
w("seeds/b.wav", synth.wav_file(np.tile(s, (1, 3)), 24, extensible=True, trailer_chunk=b"LIST\x04\x00\x00\x00abcd"))
import struct
fmt0 = struct.pack("<HHIIHH", 1, 0, 48000, 0, 0, 0)          # the all-zero fmt chunk that once divided by zero (round 1 advice)
w("seeds/c.wav", b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt0) + 8 + 16) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt0)) + fmt0 + b"data" + struct.pack("<I", 16) + bytes(16))
for k in range(6):
    w("job/seq/f_%06d.dpx" % k, synth.dpx_file(synth.components(32, 16, 3, 16, "film", seed=k), synth.PIX_RGB16_BE, frame_index=k))
w("job/a.dpx", synth.dpx_file(comp, synth.PIX_RGB16_BE)); w("job/b.dpx", synth.dpx_file(synth.components(33, 7, 3, 10, "film", seed=2), synth.PIX_RGB10_FILLEDA_BE))
w("job/a.tif", synth.tiff_file(comp, synth.PIX_RGB16_LE, trailer=b"xx")); w("job/a.exr", synth.exr_file(comp)); w("job/a.wav", synth.wav_file(s, 16)); w("job/rev", b"\x1a\x45\xdf\xa3" + bytes(60))
import json
V = json.load(open("tests/golden/vectors.json"))["ffv1_ext"]                     # records of 1-8 arbitrary table sets, transmitted transitions, coded initial states ...
for v in V:
    if v["name"] in ("ext_8sets_50x38", "ext_random_transitions_72x40", "ext_states_coded_one_of_two_40x24", "ext_rgba_3groups_48x32"):
        w("rec/%s.bin" % v["name"], open("tests/golden/" + v["config_record_file"], "rb").read() if "config_record_file" in v else bytes.fromhex(v["config_record"]))
    if v["name"] in ("ext_v1_inband_custom_48x32", "ext_v0_rgb8_48x32"):            # ... and packets with the version 0 / 1 header inside (first 600 bytes)
        w("rec/%s.bin" % v["name"], open("tests/golden/" + v["frames"][0]["packet"], "rb").read()[:600])
for i, (pixfmt, ctx, coder) in enumerate([(synth.PIX_RGB16_BE, 1, 1), (synth.PIX_RGB10_FILLEDA_BE, 0, 2), (synth.PIX_Y8, 1, 1), (synth.PIX_RGBA16_LE, 2, 2)]):
    w("rec/rec%d.bin" % i, ob.config_record(ob.Params(64, 48, pixfmt, 2, 2, 1, ctx, 0, coder, 3)))
PY
cd $R/rawcooked_amd/csrc
F="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=all -I../../include -I."
g++ $F $R/tools/fuzz/fuzz_probes.cpp rc_common.cpp formats.cpp -o $T/fuzz_probes
g++ $F $R/tools/fuzz/fuzz_record.cpp rc_common.cpp ffv1_host.cpp hashes.cpp -o $T/fuzz_record -lpthread
g++ $F $R/tools/fuzz/fuzz_argv.cpp job.cpp rc_common.cpp formats.cpp mkv_mux.cpp hashes.cpp ffv1_host.cpp -o $T/fuzz_argv -lpthread
$T/fuzz_probes $N $T/seeds/*
$T/fuzz_record $N $T/rec/*
$T/fuzz_argv $NA $T/job
