"""GPU parity for the WAV -> FLAC path: device frames == oracle frames byte for byte, decode == PCM."""
import ctypes as C

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
    enc.close()
