"""GPU parity for the WAV -> FLAC path: device frames == oracle frames byte for byte, decode == PCM."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_binding as ob
from rawcooked_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = [
    # channels, bits, rate, samples, kind
    (2, 16, 48000, 20000, "music"),
    (6, 24, 48000, 14000, "music"),      # config 3 shape: 6 ch / 24 bit / 48 kHz
    (1, 8, 44100, 9000, "music"),
    (2, 24, 96000, 20000, "noise"),
    (2, 16, 44100, 5000, "silence"),
    (1, 16, 48000, 4608 * 2 + 1, "music"),   # last block of one sample
    (2, 16, 48000, 17, "music"),             # shorter than every LPC order
    (2, 16, 48000, 20000, "stereo"),         # correlated channels: mid/side frames (channel assignment 10)
    (2, 24, 48000, 14000, "stereo_left"),    # left/side (8): the side signal is 25 bits wide
    (2, 16, 44100, 9000, "stereo_right"),    # side/right (9)
    (2, 8, 44100, 9000, "stereo"),
]


@pytest.mark.parametrize("ch,bits,rate,n,kind", CASES)
def test_flac_frames_match_oracle(built, ch, bits, rate, n, kind):
    wav = synth.wav_file(synth.pcm_samples(n, ch, bits, rate, kind), bits, rate)
    info = api.wav_probe(wav)
    pcm = wav[info.data_offset:info.data_offset + info.data_size]
    enc = api.FlacEncoder(ch, rate, bits, 0, 8)
    frames, cp = enc.encode(pcm)
    oframes, ocp = ob.flac_encode(ch, rate, bits, pcm, 0, 8)
    assert len(frames) == len(oframes)
    for i, (a, b) in enumerate(zip(frames, oframes)):
        assert a == b, f"frame {i}: device {len(a)} bytes, oracle {len(b)} bytes, first diff at {next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), -1)}"
    assert cp == ocp
    assert ob.flac_decode(ch, rate, bits, b"".join(frames), len(pcm)) == pcm
    if kind.startswith("stereo") and bits > 8:
        want = {"stereo": 10, "stereo_left": 8, "stereo_right": 9}[kind]
        assert [f[3] >> 4 for f in frames[:-1]] == [want] * (len(frames) - 1), [f[3] >> 4 for f in frames]
    enc.close()


@pytest.mark.parametrize("name", ["wav_2ch16_48k", "wav_6ch24_48k", "wav_1ch8_44k", "wav_2ch24_noise", "wav_2ch16_midside", "wav_2ch24_leftside", "wav_2ch16_sideright"])
def test_device_predictor_words_are_the_pinned_ones(built, name):
    """The device's Levinson-Durbin and quantiser (IEEE double, fixed operation order, -ffp-contract=off) choose the coefficient words
    pinned in tests/golden/flac_coefficients.json, read back out of the device's own frames by the specification-level decoder
    (tests/flac_validator.py): a rounding that moved is reported as the block, channel and coefficient it moved in.  Reference maths:
    Lib/ThirdParty/flac/src/libFLAC/lpc.c:122-259."""
    import json
    import flac_validator
    from test_validators import words_diff
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    v = next(x for x in json.load(open(os.path.join(G, "vectors.json")))["flac"] if x["name"] == name)
    pcm = open(os.path.join(G, v["pcm"]), "rb").read()
    enc = api.FlacEncoder(v["channels"], v["rate"], v["bits"], 0, 8)
    frames, cp = enc.encode(pcm)
    enc.close()
    got, infos = flac_validator.parse_stream(b"".join(frames), v["channels"], v["bits"], v["rate"])
    assert got == pcm
    words_diff(name, flac_validator.predictor_words(infos), json.load(open(os.path.join(G, "flac_coefficients.json")))[name])
    assert b"".join(frames) == open(os.path.join(G, v["frames"]), "rb").read() and cp.hex() == v["codec_private"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("RCGPU_SOAK_FLAC", "8"))))      # soak: RCGPU_SOAK_FLAC=400
def test_flac_random_signals_match_oracle(built, seed):
    """Random channel counts, depths, lengths and signals (tones + dither at random levels, steps, bursts of full-scale noise,
    silence): the predictor search runs in floating point on the device, and must pick what the scalar oracle picks."""
    import ext_streams
    import ref_decode
    ch, bits, rate, n, pcm = ext_streams.flac_signal(seed)
    enc = api.FlacEncoder(ch, rate, bits, 0, 8)
    frames, cp = enc.encode(pcm)
    oframes, ocp = ob.flac_encode(ch, rate, bits, pcm, 0, 8)
    assert frames == oframes and cp == ocp, f"seed {seed}: {ch} ch {bits} bit {n} samples"
    assert ob.flac_decode(ch, rate, bits, b"".join(frames), len(pcm)) == pcm
    if ref_decode.flac_available():                          # the reference's own FLAC wrapper and the libFLAC it ships, block by block
        assert ref_decode.flac_decode([(bits, cp, frames)]) == [(0, pcm)], f"seed {seed}: {ch} ch {bits} bit {n} samples"
    enc.close()
