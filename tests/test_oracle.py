"""CPU: the oracle against the committed golden vectors (blessed by the real reference, see golden/README.md),
against its own decoder, and -- where oracle/_ref exists -- against the real reference binary."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from rawcooked_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))


@pytest.mark.parametrize("v", VEC["ffv1"], ids=lambda v: v["name"])
def test_ffv1_golden(built, v):
    assert v["reference_check"]
    p = ob.Params(v["width"], v["height"], v["pixfmt"], v["num_h"], v["num_v"], 1, 1, v["flags"], v["coder"])
    assert ob.config_record(p).hex() == v["config_record"]
    for fr in v["frames"]:
        payload = open(os.path.join(G, fr["payload"]), "rb").read()
        packet = open(os.path.join(G, fr["packet"]), "rb").read()
        assert hashlib.sha256(payload).hexdigest() == fr["payload_sha256"]
        assert ob.encode_payload(p, payload, v["line_bytes"]) == packet          # encoder restatement
        assert ob.decode_payload(p, packet, v["line_bytes"]) == payload          # decoder restatement (FFV1_Frame/Slice.cpp)


def ext_record(v):
    return open(os.path.join(G, v["config_record_file"]), "rb").read() if "config_record_file" in v else bytes.fromhex(v["config_record"])


@pytest.mark.parametrize("v", VEC["ffv1_ext"], ids=lambda v: v["name"])
def test_ffv1_ext_golden(built, v):
    """Streams FFmpeg's defaults never produce (up to 8 table sets of arbitrary level maps, a set per plane group or per slice, a transmitted
    transition table, coded initial states, the version 0 / 1 header inside the frame): each one was checked and decoded by the real
    reference (tests/golden/make_golden.py); the oracle writes these bytes from the description and reads them back knowing only the
    picture's geometry -- parameters::Parse restated (FFV1_Parameters.cpp:23-183,206-253)."""
    assert v["reference_check"]
    e = ob.stream_ext_from_vector(v, G)
    p = ob.with_ext(ob.Params(v["width"], v["height"], v["pixfmt"], v["num_h"], v["num_v"], v["ec"]), e)
    rec = ext_record(v)
    assert ob.config_record(p) == rec and bool(rec) == (v["version"] == 3)
    for fr in v["frames"]:
        payload = open(os.path.join(G, fr["payload"]), "rb").read()
        packet = open(os.path.join(G, fr["packet"]), "rb").read()
        assert hashlib.sha256(payload).hexdigest() == fr["payload_sha256"] and hashlib.sha256(packet).hexdigest() == fr["packet_sha256"]
        assert ob.encode_payload(p, payload, v["line_bytes"]) == packet
        code, back = ob.decode_stream(ob.Params(v["width"], v["height"], v["pixfmt"]), rec, packet, v["line_bytes"])
        assert code == 0 and back == payload
        bad = bytearray(packet)
        bad[len(bad) // 3] ^= 0x40                      # (not the middle: with four slices of noise that is the slack at the end of a slice)
        code, back = ob.decode_stream(ob.Params(v["width"], v["height"], v["pixfmt"]), rec, bytes(bad), v["line_bytes"])
        assert code != 0 or back != payload


def test_random_streams_of_the_general_syntax_are_what_the_reference_decodes(built):
    """The oracle's WRITER of the general syntax, stream by stream against the real reference's decoder: 150 random descriptions (tests/
    ext_streams.py: 1 to 8 random table sets, a set per plane group, transition tables, coded initial states, version 1 headers, six pixel
    layouts, random geometry and slices), three pictures each, handed to ffv1_frame::OutOfBand / ::Process by oracle/ref_ffv1_decode.cpp --
    every one decodes to its source without a complaint, and the oracle's own reader agrees.  (Where the driver was not built -- no
    /root/reference -- the thirteen blessed vectors above carry the claim alone; the GPU test of the same streams checks the device.)"""
    import ext_streams
    import ref_decode
    if not ref_decode.available():
        pytest.skip("oracle/_ref/ref_ffv1_decode not built (needs /root/reference)")
    n = int(os.environ.get("RCGPU_SOAK_EXT_CPU", "150"))
    streams = [ext_streams.random_stream(seed) for seed in range(n)]
    res, lines = ref_decode.decode([(fl, flags, w, h, rec, pks) for w, h, _, _, rec, pks, _, fl, flags, _ in streams])
    for (w, h, pixfmt, line_bytes, rec, pks, pls, _, _, tight), frames, line in zip(streams, res, lines):
        assert ext_streams.reference_decodes_to(frames, tight), line
        code, back = ob.decode_stream(ob.Params(w, h, pixfmt), rec, pks[-1], line_bytes)
        assert code == 0 and back == pls[-1], line
    # and the reference's complaint when a stream is NOT what it says: a flipped bit in the middle of a frame
    w, h, _, _, rec, pks, _, fl, flags, tight = streams[0]
    bad = bytearray(pks[0]); bad[len(bad) // 3] ^= 0x10
    (frames,), _ = ref_decode.decode([(fl, flags, w, h, rec, [bytes(bad)])])
    assert frames[0][0] != 0 or frames[0][1][:len(tight[0])] != tight[0]


def test_default_streams_of_random_geometry_are_what_the_reference_decodes(built):
    """The oracle's ENCODER -- whose bytes the device encoder must equal -- against the real reference's decoder, beyond the 33 blessed vectors:
    the pictures of the GPU soak (tests/ext_streams.py::mixed_content_stream: flat patches, ramps, noise; random sizes and slice grids; twelve
    pixel layouts), both context models, handed to ffv1_frame::OutOfBand / ::Process by oracle/ref_ffv1_decode.cpp: every frame decodes to
    its picture without a complaint."""
    import ext_streams
    import ref_decode
    if not ref_decode.available():
        pytest.skip("oracle/_ref/ref_ffv1_decode not built (needs /root/reference)")
    cases, want = [], []
    for seed in range(int(os.environ.get("RCGPU_SOAK_SEEDS_CPU", "72"))):
        m = ext_streams.mixed_content_stream(seed)
        if not m["reference_takes_it"]:
            continue
        for ctx in (1, 2):
            p = ob.Params(m["w"], m["h"], m["pixfmt"], m["nh"], m["nv"], 1, ctx)
            cases.append((m["flavor"], 0, m["w"], m["h"], ob.config_record(p), [ob.encode_payload(p, pl, m["line_bytes"]) for pl in m["payloads"]]))
            want.append(m["tight"])
    assert len(cases) >= 80
    res, lines = ref_decode.decode(cases)
    for frames, tight, line in zip(res, want, lines):
        assert ext_streams.reference_decodes_to(frames, tight), line


def test_verdicts_on_damaged_frames_are_the_reference_s(built):
    """The oracle's decoder is what the GPU tests hold the device's complaints to ("flagged, or the source's bytes").  Here it is held to the
    real reference's decoder itself on damaged frames -- one flipped bit anywhere in a frame, a third of them in its last 40 bytes (slice
    sizes, CRCs), with and without slice CRCs, four pixel layouts: the same frames draw a complaint from both, and the frames both decode
    quietly decode to the same bytes (oracle/_ref/ref_ffv1_decode, a decoder per frame: its first complaint stays with it)."""
    import ext_streams
    import ref_decode
    from rawcooked_amd import synth
    if not ref_decode.available():
        pytest.skip("oracle/_ref/ref_ffv1_decode not built (needs /root/reference)")
    quiet = loud = 0
    for pixfmt, ctx, ec in [(synth.PIX_RGB16_BE, 1, 0), (synth.PIX_RGB16_BE, 1, 1), (synth.PIX_RGB10_FILLEDA_BE, 1, 0), (synth.PIX_Y16_BE, 2, 0), (synth.PIX_RGBA16_LE, 1, 1)]:
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        w, h, nh, nv = 48, 32, 2, 2
        comp = synth.components(w, h, nc, bits, "film", seed=9)
        payload, line_bytes = synth.pack_payload(comp, pixfmt, True)
        tight = synth.pack_payload(comp, pixfmt, False)[0]
        tl = len(tight) // h
        p = ob.Params(w, h, pixfmt, nh, nv, ec, ctx)
        good, rec = ob.encode_payload(p, payload, line_bytes), ob.config_record(p)
        rng = np.random.default_rng(3)
        variants = []
        for k in range(int(os.environ.get("RCGPU_SOAK_VERDICTS", "120"))):
            b = bytearray(good)
            at = int(rng.integers(0, len(b))) if k % 3 else len(b) - 1 - int(rng.integers(0, 40))
            b[at] ^= 1 << int(rng.integers(0, 8))
            variants.append(bytes(b))
        res, _ = ref_decode.decode([(ext_streams.flavor_of(pixfmt), 0, w, h, rec, [pk]) for pk in variants])
        for (frame,), pk in zip(res, variants):
            verdict, ref_bytes = frame
            code, back = ob.decode_stream(ob.Params(w, h, pixfmt), rec, pk, line_bytes)
            assert (verdict == 0) == (code == 0), (pixfmt, ec, verdict, code)
            if code == 0:
                assert ref_bytes[:len(tight)] == b"".join(back[y * line_bytes:y * line_bytes + tl] for y in range(h))
                quiet += 1
            else:
                loud += 1
    assert loud > 500 and quiet >= 1


def _muxed_package_checks(refbin, name, w, h, pixfmt, n, fps, audio=None, attach=(), kind="film"):
    """A package (DPX sequence, optional WAV, optional attachments) analysed by the real reference (`-d`: its reversibility data, its -slices),
    encoded by the oracle, muxed by the PRODUCT's Matroska writer (api.MkvMuxer, host code: no device), checked by the real reference."""
    import shutil
    import subprocess
    import tempfile
    from rawcooked_amd import api, synth

    from ref_decode import run_reference as run          # (retried: the reference's thread pool loses a wake-up at shutdown now and then)
    work = tempfile.mkdtemp()
    try:
        os.makedirs(work + "/pkg/img")
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        comps = [synth.components(w, h, nc, bits, kind, seed=s) for s in range(min(n, 7))]
        files = []
        for i in range(n):
            fn = work + "/pkg/img/f_%06d.dpx" % i
            open(fn, "wb").write(synth.dpx_file(comps[i % len(comps)], pixfmt, frame_index=i))
            files.append(fn)
        if audio:
            ch, abits, rate, ns = audio
            wav = synth.wav_file(np.random.default_rng(5).integers(-2000, 2000, size=(ns, ch)).astype(np.int32), abits, rate)
            open(work + "/pkg/snd.wav", "wb").write(wav)
        for an, ad in attach:
            open(work + "/pkg/" + an, "wb").write(ad)
        r = run([refbin, "--hash", "--no-check-padding", "--check", "-d", "-y", "-framerate", "%d/%d" % fps if fps[1] != 1 else str(fps[0]), "pkg"], work)
        assert r.returncode == 0, (name, r.stdout[-500:], r.stderr[-500:])
        info = api.dpx_probe(open(files[0], "rb").read())
        nh, nv = api.slices_to_grid(int(r.stdout.split("-slices ")[1].split()[0]))
        p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
        mux = api.MkvMuxer(work + "/pkg.mkv")
        tv = mux.add_video(ob.config_record(p), w, h, fps[0], fps[1])
        blocks, cache = [], {}
        for i, fn in enumerate(files):
            k = i % len(comps)
            if k not in cache:
                b = open(fn, "rb").read()
                cache[k] = ob.encode_payload(p, b[info.data_offset:info.data_offset + info.data_size], info.line_bytes)
            blocks.append((i * 1000000000 * fps[1] // fps[0], 0, tv, cache[k]))
        if audio:
            winfo = api.wav_probe(wav)
            frames, cp = ob.flac_encode(ch, rate, abits, wav[winfo.data_offset:winfo.data_offset + winfo.data_size], 0, 8)
            ta = mux.add_audio(cp, ch, rate, abits)
            for k, fr in enumerate(frames):
                blocks.append((k * 4608 * 1000000000 // rate, 1, ta, fr))
        for fn in sorted(os.listdir(work + "/pkg")):
            if fn not in ("img", "snd.wav"):
                mux.add_attachment("pkg/" + fn, open(work + "/pkg/" + fn, "rb").read())
        mux.add_attachment("RAWcooked reversibility data", open(work + "/pkg.rawcooked_reversibility_data", "rb").read())
        mux.begin()
        for pts, _, trk, data in sorted(blocks, key=lambda x: (x[0], x[1])):
            mux.write_block(trk, pts, data)
        mux.close()
        r = run([refbin, "--check", "pkg.mkv"], work)
        assert r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout, (name, r.stdout[-800:], r.stderr[-800:])
    finally:
        shutil.rmtree(work)


@pytest.mark.parametrize("name,args", [
    ("long: 2500 frames, 104 s of timestamps, many clusters", dict(w=16, h=8, pixfmt=0, n=2500, fps=(24, 1))),
    ("24000/1001", dict(w=32, h=16, pixfmt=5, n=120, fps=(24000, 1001))),
    ("one frame per second", dict(w=32, h=16, pixfmt=1, n=40, fps=(1, 1))),
    ("60 fps beside 6 channels of 24-bit FLAC", dict(w=48, h=32, pixfmt=5, n=90, fps=(60, 1), audio=(6, 24, 48000, 72000))),
    ("audio outlasting the pictures", dict(w=32, h=16, pixfmt=11, n=10, fps=(24, 1), audio=(2, 16, 44100, 200000))),
    ("attachments: text, 300 KB, empty", dict(w=32, h=16, pixfmt=9, n=12, fps=(25, 1), attach=[("notes.txt", b"hello"), ("big.bin", bytes(range(256)) * 1200), ("empty.dat", b"")])),
    ("frames of 19 MB: four-byte EBML sizes", dict(w=2048, h=1556, pixfmt=5, n=2, fps=(24, 1), kind="noise")),
], ids=lambda x: x if isinstance(x, str) else "")
def test_reference_checks_what_the_muxer_writes(built, refbin, name, args):
    """The product's Matroska writer (mkv_mux.cpp) beyond the shapes the GPU end-to-end tests give it, judged by the real reference's demuxer
    and --check on the CPU: long timestamp ranges, rational and slow frame rates, audio interleaved and outlasting the video, attachments
    of every size, blocks beyond 16 MB."""
    from rawcooked_amd import synth
    pf = {0: synth.PIX_RGB8, 1: synth.PIX_RGB10_FILLEDA_BE, 5: synth.PIX_RGB16_BE, 9: synth.PIX_RGBA16_LE, 11: synth.PIX_Y16_BE}[args.pop("pixfmt")]
    _muxed_package_checks(refbin, name, pixfmt=pf, **args)


def reference_misdecodes(width, num_h, pixels_per_block):
    """The rule of oracle/route_c_ffv1_frame_cpp.patch (RefMisdecodes), restated: the reference's decoder merges the blocks two slices share
    only when slice (0,0) itself ends inside a block (Lib/Transform/Transform.cpp:176-177,733-734,871-872) and only two slices to a block."""
    if num_h <= 1 or pixels_per_block <= 1:
        return False
    edge = [i * width // num_h for i in range(num_h + 1)]
    other = any(edge[i] % pixels_per_block for i in range(2, num_h))
    narrow = any(edge[i + 1] - edge[i] < pixels_per_block and (edge[i] % pixels_per_block or edge[i + 1] % pixels_per_block) for i in range(num_h))
    return other and (edge[1] % pixels_per_block == 0 or narrow)


def test_the_geometries_the_reference_rebuilds_wrongly_are_the_ones_route_c_leaves_to_it(built):
    """12-bit packed RGB / Y and 10-bit filled Y pack several pixels into a block, and slices may start inside one.  The unmodified reference
    rebuilds such a picture wrongly when its first slice ends ON a block and others do not (146 pixels in 6 columns: its own --check of a
    file it agreed to encode fails) -- the device decoder rebuilds it right, and route C must not pass what the reference fails.  The rule
    the hook uses is held to the reference's decoder here: wherever the rule says "the reference is right" the reference's decoder returns
    the source's bytes (never the other way round), and the textbook case of the rule really is rebuilt wrongly."""
    import ref_decode
    from rawcooked_amd import api, synth
    if not ref_decode.available():
        pytest.skip("oracle/_ref/ref_ffv1_decode not built (needs /root/reference)")
    ppb = {synth.PIX_RGB12_PACKED_BE: 8, synth.PIX_Y10_FILLEDA_BE: 3, synth.PIX_Y10_FILLEDB_BE: 3, synth.PIX_Y12_PACKED_BE: 8}
    right = wrong = 0
    for seed in range(int(os.environ.get("RCGPU_SOAK_SPAN", "160"))):
        rng = np.random.default_rng(900 + seed)
        pixfmt = list(ppb)[seed % 4]
        flags = [0, 1][seed % 2] if ppb[pixfmt] == 8 else 0
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        w, h = int(rng.integers(16, 200)), int(rng.integers(6, 40))
        try:
            f = synth.dpx_file(synth.components(w, h, nc, bits, "noise", seed=seed), pixfmt, flags=flags)
            info = api.dpx_probe(f)
        except RuntimeError:
            continue
        nh, nv = api.slices_to_grid(info.slices if seed % 5 else [4, 6, 9, 12, 16][seed % 5])
        if nh >= w or nv >= h:
            continue
        payload = f[info.data_offset:info.data_offset + info.data_size]
        p = ob.Params(w, h, pixfmt, nh, nv, 1, 1, flags)
        try:
            (frames,), _ = ref_decode.decode([(info.flavor.decode(), flags, w, h, ob.config_record(p), [ob.encode_payload(p, payload, info.line_bytes)])])
            ok = frames[0][0] == 0 and frames[0][1][:len(payload)] == payload
        except AssertionError:
            ok = False                                       # (the reference's decoder did not survive the geometry)
        if not reference_misdecodes(w, nh, ppb[pixfmt]):
            assert ok, (seed, w, h, nh, nv, pixfmt, flags)
            right += 1
        elif not ok:
            wrong += 1
    assert right > 80 and wrong >= 5
    (frames,), _ = ref_decode.decode([_span_case(146, 45, synth.PIX_RGB12_PACKED_BE)])
    assert reference_misdecodes(146, 6, 8) and frames[0][0] == 0 and frames[0][1] != _span_case.payload


def _span_case(w, h, pixfmt):
    from rawcooked_amd import api, synth
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    f = synth.dpx_file(synth.components(w, h, nc, bits, "film", seed=0), pixfmt)
    info = api.dpx_probe(f)
    nh, nv = api.slices_to_grid(info.slices)
    _span_case.payload = f[info.data_offset:info.data_offset + info.data_size]
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    return info.flavor.decode(), 0, w, h, ob.config_record(p), [ob.encode_payload(p, _span_case.payload, info.line_bytes)]


def test_flac_of_random_signals_is_what_the_reference_decodes(built):
    """The oracle's FLAC ENCODER -- whose frames the device encoder must equal -- against the reference's own FLAC path (its wrapper around the
    libFLAC it ships, fed CodecPrivate and blocks the way track_info feeds them; oracle/ref_flac_decode.cpp), beyond the 7 blessed streams:
    the random signals of the GPU soak (tests/ext_streams.py::flac_signal: 1 to 8 channels, 8 / 16 / 24 bit, three rates, 1 to 13 874
    samples) decode to their PCM bytes without a complaint."""
    import ext_streams
    import ref_decode
    if not ref_decode.flac_available():
        pytest.skip("oracle/_ref/ref_flac_decode not built (needs /root/reference)")
    cases, want = [], []
    for seed in range(int(os.environ.get("RCGPU_SOAK_FLAC_CPU", "40"))):
        ch, bits, rate, n, pcm = ext_streams.flac_signal(seed)
        frames, cp = ob.flac_encode(ch, rate, bits, pcm, 0, 8)
        cases.append((bits, cp, frames)); want.append(pcm)
    assert ref_decode.flac_decode(cases) == [(0, p) for p in want]


@pytest.mark.parametrize("v", VEC["flac"], ids=lambda v: v["name"])
def test_flac_golden(built, v):
    pcm = open(os.path.join(G, v["pcm"]), "rb").read()
    frames = open(os.path.join(G, v["frames"]), "rb").read()
    got, cp = ob.flac_encode(v["channels"], v["rate"], v["bits"], pcm, 0, 8)
    assert b"".join(got) == frames and [len(f) for f in got] == v["frame_sizes"] and cp.hex() == v["codec_private"]
    assert ob.flac_decode(v["channels"], v["rate"], v["bits"], frames, len(pcm)) == pcm


def test_config_record_parses_and_crc_is_zero(built):
    for pixfmt in range(13):
        p = ob.Params(64, 48, pixfmt, 3, 2, 1, 1)
        rec = ob.config_record(p)
        assert ob.lib().ffv1o_crc32(rec, len(rec)) == 0                          # FFV1_Frame.cpp:116
        q = ob.Params(64, 48, pixfmt, 0, 0, 0, 1)
        assert ob.lib().ffv1o_parse_config_record(rec, len(rec), __import__("ctypes").byref(q)) == 0
        assert (q.num_h_slices, q.num_v_slices, q.ec) == (3, 2, 1)


EDGE = [  # w, h, pixfmt, nh, nv, kind  -- ragged slices, 1x1 grid, single-column slices, every layout
    (7, 5, synth.PIX_RGB16_BE, 1, 1, "noise"), (65, 33, synth.PIX_RGB10_FILLEDA_LE, 4, 3, "film"), (9, 9, synth.PIX_RGB8, 4, 4, "noise"),
    (31, 8, synth.PIX_RGBA8, 5, 3, "film"), (20, 20, synth.PIX_RGBA16_LE, 3, 3, "noise"), (40, 13, synth.PIX_Y16_LE, 4, 2, "noise"),
    (40, 13, synth.PIX_Y8, 2, 2, "flat"), (16, 16, synth.PIX_RGB12_FILLEDA_LE, 2, 2, "film"), (128, 4, synth.PIX_RGB16_LE, 3, 2, "flat"),
]


@pytest.mark.parametrize("w,h,pixfmt,nh,nv,kind", EDGE)
def test_roundtrip_edges(built, w, h, pixfmt, nh, nv, kind):
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    payload, lb = synth.pack_payload(synth.components(w, h, nc, bits, kind, seed=w * 131 + h), pixfmt, True)
    for ec in (0, 1):
        for ctx in (0, 1):
            p = ob.Params(w, h, pixfmt, nh, nv, ec, ctx)
            pkt = ob.encode_payload(p, payload, lb)
            assert ob.decode_payload(p, pkt, lb) == payload
    bad = bytearray(pkt)
    bad[len(bad) // 2] ^= 1
    out = __import__("ctypes").create_string_buffer(lb * h)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    assert ob.lib().ffv1o_decode_payload(__import__("ctypes").byref(p), bytes(bad), len(bad), out, lb) != 0   # slice CRC catches it


def _run_ref(cmd, cwd, timeout=120, attempts=2):
    """One retry (idempotent steps, -y): the reference's third-party thread pool can lose its shutdown wake-up (ThreadPool.h:27-33 against
    :66-68; matroska::Shutdown() then never returns from std::thread::join, Matroska.cpp:271-274); stacks in profiles/r04_hang_stacks.txt."""
    for attempt in range(attempts):
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL)
        except subprocess.TimeoutExpired:
            if attempt + 1 == attempts:
                raise


def test_oracle_packets_pass_the_real_reference(built, refbin, tmp_path):
    """Re-pins the oracle here and now (not only through the committed vectors): DPX 16-bit + WAV package."""
    work = str(tmp_path)
    os.makedirs(work + "/pkg/img")
    w, h, pixfmt = 72, 40, synth.PIX_RGB16_BE
    files = []
    for i in range(2):
        fn = work + "/pkg/img/f_%06d.dpx" % i
        open(fn, "wb").write(synth.dpx_file(synth.components(w, h, 3, 16, "film", seed=i), pixfmt, frame_index=i))
        files.append(fn)
    wav = synth.wav_file(synth.pcm_samples(5000, 2, 16, 48000), 16, 48000)
    open(work + "/pkg/snd.wav", "wb").write(wav)
    r = _run_ref([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0, r.stderr
    info = api.dpx_probe(open(files[0], "rb").read())
    nh, nv = api.slices_to_grid(info.slices)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    ai = api.wav_probe(wav)
    pcm = wav[ai.data_offset:ai.data_offset + ai.data_size]
    frames, cp = ob.flac_encode(2, 48000, 16, pcm, 0, 8)
    mux = api.MkvMuxer(work + "/pkg.mkv")
    tv = mux.add_video(ob.config_record(p), w, h, 24, 1)
    ta = mux.add_audio(cp, 2, 48000, 16)
    mux.add_attachment("RAWcooked reversibility data", open(work + "/pkg.rawcooked_reversibility_data", "rb").read())
    mux.begin()
    B = int.from_bytes(cp[8:10], "big")
    for i, f in enumerate(frames):
        mux.write_block(ta, i * B * 10 ** 9 // 48000, f)
    for i, fn in enumerate(files):
        b = open(fn, "rb").read()
        mux.write_block(tv, i * 10 ** 9 // 24, ob.encode_payload(p, b[info.data_offset:info.data_offset + info.data_size], info.line_bytes))
    mux.close()
    r = _run_ref([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout, r.stdout + r.stderr


def _padding_vectors():
    return json.load(open(os.path.join(G, "padding_vectors.json")))["vectors"]


def _padding_payload(v):
    bits, nc, _, _ = synth.PIX_INFO[v["pixfmt"]]
    pl, _ = synth.pack_payload(synth.components(v["width"], v["height"], nc, bits, "noise", seed=v["seed"]), v["pixfmt"], True, v["flags"])
    b = bytearray(len(pl)) if v["zero"] else bytearray(pl)
    for at, x in v["pokes"]:
        b[at] ^= x
    return bytes(b)


def test_padding_oracle_matches_the_reference_parser(built):
    """oracle/dpx_oracle.c against tests/golden/padding_vectors.json: In_FirstNonZero as the REAL dpx::ParseBuffer computed it
    (DPX.cpp:501-608; made observable by oracle/ref_padding_probe.cpp, see tests/golden/make_padding_golden.py)."""
    vs = _padding_vectors()
    assert len(vs) > 600 and sum(1 for v in vs if v["first_nonzero"] >= 0) > 100
    for v in vs:
        bits, nc, _, be = synth.PIX_INFO[v["pixfmt"]]
        packing = synth.DPX_PACKING.get(v["pixfmt"], 1 if bits in (10, 12) else 0)
        got = ob.dpx_padding_first_nonzero(_padding_payload(v), v["width"], v["height"], bits, nc, be, packing, bool(v["flags"] & synth.FLAG_ALTERN))
        want = 2 ** 64 - 1 if v["first_nonzero"] < 0 else v["first_nonzero"]
        assert got == want, (v["flavor"], v["width"], v["height"], v["flags"], v["pokes"], got, want)
