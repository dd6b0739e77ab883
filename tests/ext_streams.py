"""Random FFV1 streams of the general syntax -- what parameters::Parse accepts and FFmpeg's defaults never produce -- written by the oracle
(whose writer of this syntax is pinned by the reference-blessed ext vectors, and stream by stream by oracle/_ref/ref_ffv1_decode where it
exists).  Shared by the CPU test (oracle writer against the real reference's decoder) and the GPU test (device decoder)."""
import ctypes

import numpy as np

import oracle_binding as ob
from rawcooked_amd import api, synth


def random_ext(rng, bits):
    """a random stream description within what parameters::Parse accepts: 1..8 table sets, each five random level maps whose context count
    stays small enough for a test, a random set per plane group, now and then a transition table of its own and coded initial states"""
    def table(max_levels):
        L = int(rng.integers(1, max_levels + 1))
        if L == 1:
            return [128]
        cuts = sorted(rng.choice(np.arange(1, 128), size=L - 1, replace=False).tolist())
        return [b - a for a, b in zip([0] + cuts, cuts + [128])]
    sets = []
    for _ in range(int(rng.integers(1, 9))):
        while True:
            t = [table(6), table(5), table(4), table(3) if rng.random() < 0.6 else [128], table(3) if rng.random() < 0.6 else [128]]
            n = 1
            for r in t:
                n *= 2 * len(r) - 1
            if n <= 6000:
                break
        sets.append(t)
    kw = dict(sets=sets, set_index=tuple(int(rng.integers(0, len(sets))) for _ in range(3)))
    custom = rng.random() < 0.5
    if custom:
        kw["one_state"] = [0] + [int(min(255, max(1, i + rng.integers(1, 20) - (rng.random() < 0.1) * rng.integers(0, 10)))) for i in range(1, 256)]
    return kw, custom



def random_stream(seed, n=3):
    """(width, height, pixfmt, line_bytes, record, packets, payloads, flavor string, flags) of seed's stream: geometry, pixel layout, table
    sets, transitions, coded initial states, version 1 or 3 all drawn from it."""
    rng = np.random.default_rng(9000 + seed)
    pixfmt = [synth.PIX_RGB16_BE, synth.PIX_RGB10_FILLEDA_LE, synth.PIX_RGBA16_LE, synth.PIX_Y16_BE, synth.PIX_RGB8, synth.PIX_RGB12_PACKED_BE][seed % 6]
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    w, h = int(rng.integers(40, 120)), int(rng.integers(24, 72))
    if pixfmt == synth.PIX_RGB12_PACKED_BE:
        w = 96                                               # (a geometry the reference's word merging handles, tests/golden/make_golden.py)
    nh = int(rng.integers(1, 5)); nv = int(rng.integers(1, nh + 1))
    kw, custom = random_ext(rng, bits)
    version = 1 if seed % 5 == 4 else 3
    if version == 1:
        kw = dict(version=1, sets=kw["sets"][:1], **({"one_state": kw["one_state"]} if custom else {}))
        nh = nv = 1
    e = ob.stream_ext(**kw)
    if version == 3 and rng.random() < 0.5:                    # coded initial states for the sets in use: values from which state 0 is out of reach
        lo, hi = (1, 255) if custom else (30, 226)
        used = sorted(set(kw["set_index"][:(nc - 1 if nc != 1 else 1)]))
        init = {i: bytes(rng.integers(lo, hi + 1, size=ob.lib().ffv1o_ext_context_count(ctypes.byref(e), i) * 32, dtype=np.int64).astype(np.uint8)) for i in used}
        e = ob.stream_ext(initial_states=init, **kw)
    ec = int(rng.integers(0, 2)) if version == 3 else 0
    p = ob.with_ext(ob.Params(w, h, pixfmt, nh, nv, ec), e)
    pls, tight = [], []
    for i in range(n):
        comp = synth.components(w, h, nc, bits, ["film", "noise", "flat"][(seed + i) % 3], seed=seed * 10 + i)
        pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
        pls.append(pl)
        tight.append(synth.pack_payload(comp, pixfmt, False)[0])
    rec = ob.config_record(p)
    pks = [ob.encode_payload(p, pl, line_bytes) for pl in pls]
    flavor = api.dpx_probe(synth.dpx_file(comp, pixfmt)).flavor.decode()
    return w, h, pixfmt, line_bytes, rec, pks, pls, flavor, 0, tight


def reference_decodes_to(frames, tight):
    """What oracle/_ref/ref_ffv1_decode returned for a stream against its source pictures.  The reference's decoder lays the lines of the 8 and
    16 bit DPX flavors out one behind the other (raw_frame::DPX_Create, RawFrame.cpp:89-113: no room at a line's end, the whole plane rounded up
    to 32 bits) -- its PARSER expects every line rounded up to 32 bits (DPX.cpp:476-482), which is what `payloads` are; `tight` is the same
    picture without that room.  Route C takes line_bytes from the reference's plane, so the device writes what the reference would have."""
    return len(frames) == len(tight) and all(v == 0 and len(got) - len(t) in (0, 1, 2, 3) and got[:len(t)] == t for (v, got), t in zip(frames, tight))


_FLAVOR = {}


def flavor_of(pixfmt):
    """the reference's name of a pixel layout (the string its --info prints and its reversibility data carries), via the DPX probe"""
    if pixfmt not in _FLAVOR:
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        _FLAVOR[pixfmt] = api.dpx_probe(synth.dpx_file(synth.components(96, 8, nc, bits, "flat", seed=1), pixfmt)).flavor.decode()
    return _FLAVOR[pixfmt]


def mixed_content_stream(seed):
    """The pictures of tests/test_gpu_stages.py::test_random_geometries_mixed_content: flat patches, ramps and noise at random sizes, slice grids,
    segment counts and range-coder mappings, three per seed, in twelve pixel layouts.  Returns a dict; `reference_takes_it` says whether the
    real reference's decoder can be asked about this geometry: with several pixels to a block (the bit-packed flavors) its parser refuses
    pictures whose slices would start inside a block (DPX.cpp:443-456), and its decoder, shown one all the same, writes out of bounds --
    there the oracle and the device decoder stand alone."""
    rng = np.random.default_rng(1000 + seed)
    pixfmt = [synth.PIX_RGB16_BE, synth.PIX_RGB10_FILLEDA_BE, synth.PIX_RGB8, synth.PIX_Y16_LE, synth.PIX_RGBA16_LE, synth.PIX_Y8,
              synth.PIX_RGB12_PACKED_BE, synth.PIX_Y10_FILLEDA_BE, synth.PIX_RGBA10_FILLEDA_LE, synth.PIX_RGBA12_PACKED_BE, synth.PIX_Y12_PACKED_BE,
              synth.PIX_RGB12_FILLEDA_LE][seed % 12]
    bits, nc, bpp, _ = synth.PIX_INFO[pixfmt]
    w, h = int(rng.integers(24, 200)), int(rng.integers(10, 120))
    slices = [1, 4, 6, 9, 12][int(rng.integers(0, 5))]
    nh, nv = api.slices_to_grid(slices)
    if nh >= w or nv >= h:
        nh = nv = 1
    segments = [0, 1, 3, 7][int(rng.integers(0, 4))]
    rc_span = [0, 1, 8, 9, 31, 64][int(rng.integers(0, 6))]      # range-coder mapping: automatic, whole slices, split into spans of N pieces
    maxv = (1 << bits) - 1
    payloads, tight = [], []
    for f in range(3):
        comp = np.zeros((h, w, nc), dtype=np.uint16)
        for _ in range(12):                                   # random patches
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            y1, x1 = int(rng.integers(y0, h)) + 1, int(rng.integers(x0, w)) + 1
            kind = int(rng.integers(0, 3))
            if kind == 0:
                comp[y0:y1, x0:x1] = rng.integers(0, maxv + 1, size=nc)
            elif kind == 1:
                ramp = (np.arange(x1 - x0)[None, :, None] * int(rng.integers(1, 4)) + np.arange(y1 - y0)[:, None, None] + int(rng.integers(0, maxv // 2))) & maxv
                comp[y0:y1, x0:x1] = ramp
            else:
                comp[y0:y1, x0:x1] = rng.integers(0, maxv + 1, size=(y1 - y0, x1 - x0, nc))
        pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
        payloads.append(pl)
        tight.append(synth.pack_payload(comp, pixfmt, False)[0])
    takes = True
    if bpp == 0:
        takes = nh == 1
        if takes:
            try:
                api.dpx_probe(synth.dpx_file(comp, pixfmt))
            except RuntimeError:
                takes = False
    return dict(w=w, h=h, pixfmt=pixfmt, nh=nh, nv=nv, segments=segments, rc_span=rc_span, payloads=payloads, tight=tight, line_bytes=line_bytes,
                flavor=flavor_of(pixfmt), reference_takes_it=takes)


def flac_signal(seed):
    """(channels, bits, rate, samples, WAV-style PCM bytes) of tests/test_gpu_flac.py's random signals: tones + dither at random levels, steps,
    bursts of full-scale noise, silence; now and then a correlated stereo pair."""
    rng = np.random.default_rng(4000 + seed)
    ch = int([1, 2, 2, 4, 6, 8][int(rng.integers(0, 6))]); bits = int([8, 16, 16, 24, 24][int(rng.integers(0, 5))]); rate = int([44100, 48000, 96000][int(rng.integers(0, 3))])
    n = int(rng.integers(1, 3 * 4608 + 50))
    full = (1 << (bits - 1)) - 1
    t = np.arange(n) / rate
    pcm_i = np.zeros((n, ch), dtype=np.int64)
    for c in range(ch):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            sig = np.zeros(n)
        elif kind == 1:
            sig = rng.uniform(0.001, 1.0) * np.sin(2 * np.pi * rng.uniform(20, 15000) * t + rng.uniform(0, 6)) + rng.uniform(0, 0.05) * (rng.random(n) - 0.5)
        elif kind == 2:
            sig = rng.uniform(-1, 1, size=n) * rng.uniform(0.0, 1.0)
        elif kind == 3:
            sig = np.repeat(rng.uniform(-1, 1, size=n // 97 + 1), 97)[:n]                     # steps
        else:
            sig = 0.3 * np.sin(2 * np.pi * 440 * t); a = int(rng.integers(0, n)); sig[a:a + int(rng.integers(1, 600))] = rng.uniform(-1, 1)
        pcm_i[:, c] = np.clip(np.round(sig * full), -full - 1, full)
    if ch == 2 and seed % 2:           # a correlated pair: the stereo assignments (left/side, side/right, mid/side) get their turn
        w = rng.uniform(0.0, 0.2)
        pcm_i[:, 1] = np.clip(pcm_i[:, 0] * rng.choice([1.0, -1.0, 0.5]) + np.round(w * pcm_i[:, 1]), -full - 1, full)
    wav = synth.wav_file(pcm_i.astype(np.int32), bits, rate)
    info = api.wav_probe(wav)
    pcm = wav[info.data_offset:info.data_offset + info.data_size]
    return ch, bits, rate, n, pcm
