"""CPU: host-side product logic -- C-ABI surface, probes, slice rules (reference's golden table), muxer layout,
hashes, and that the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import hashlib
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from rawcooked_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rcgpu.h")).read()
    declared = set(re.findall(r"\b(rcgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    L = api.lib()
    for name in declared:
        assert hasattr(L, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, nm), f"{name} not exported"
    assert b"rcgpu version" in L.rcgpu_version()


def test_device_bytes_per_frame_needs_no_device(built):
    """rcgpu_ffv1_device_bytes_per_frame: what one frame in flight costs on the device, for callers that size max_batch (the job level does).
    BASELINE config 2 (4096x2160 RGB16, 64 slices) in run-on mode: 506 MB at the worst-case decision count (35 per sample; film content: ~460),
    against 700 before round 6 laid the slices' coded bytes into the dead front of the symbol buffer; RCGPU_FLAG_OWN_SLICE_BUFFERS and
    slices too small for that (576 slices at 4K) pay for their byte buffers."""
    import ctypes as C
    L = api.lib()
    def per(w, h, nh, nv, flags=0, run_on=1):
        cfg = api.Ffv1Config(w, h, synth.PIX_RGB16_BE, w * 6, nh, nv, 1, 1, 336, 0, 0, flags, 1, 3, 0, 0)
        return L.rcgpu_ffv1_device_bytes_per_frame(C.byref(cfg), run_on)
    raw = 4096 * 2160 * 6
    a, b = per(4096, 2160, 8, 8), per(4096, 2160, 8, 8, 0x100)
    assert 440e6 < a < 520e6 and b - a >= 2 * (raw * 3 // 2) and per(4096, 2160, 8, 8, run_on=0) < a - 100e6
    assert per(4096, 2160, 32, 18) == per(4096, 2160, 32, 18, 0x100)            # 128 x 120 slices (46 K samples): buffers of their own either way
    assert per(8192, 4320, 32, 18) < per(8192, 4320, 32, 18, 0x100) - 900e6     # config 4's shape: 256 x 240 slices, the overlay saves two sets of 468 MB
    assert L.rcgpu_ffv1_device_bytes_per_frame(None, 1) == 0


def test_product_does_not_link_the_oracle(built):
    ldd = subprocess.run(["ldd", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd
    for fn in os.listdir(os.path.join(ROOT, "rawcooked_amd", "csrc")):
        if fn.endswith((".cpp", ".hip", ".h")):
            assert "oracle" not in open(os.path.join(ROOT, "rawcooked_amd", "csrc", fn)).read().replace("oracle/flac_oracle.c", "").replace("scalar oracle", "").replace("the oracle", "").replace("oracle's", "").replace("oracle ", "")


def test_shipped_library_has_no_switch_that_changes_bytes(built):
    """The measuring switches -- kernels skipped, coder mappings forced, device aliases -- exist in the timing build only (`make timing`,
    -DRCGPU_TIMING_BUILD).  The shipped library and shim read no environment variable that can change a byte of the output: the
    reference's only such switches are its command line's (CLI/Global.cpp:938-989)."""
    csrc = os.path.join(ROOT, "rawcooked_amd", "csrc")
    allowed = {"RCGPU_BATCH", "RCGPU_DEVICES", "RCGPU_LANES", "RCGPU_MKV_MMAP", "RCGPU_MKV_NO_MMAP", "RCGPU_READERS", "RCGPU_WRITERS",
               "RCGPU_NO_CU_PARTITION", "RCGPU_RELEASE_AT_EXIT", "RCGPU_TRACE", "RCGPU_TRACE_KEPT", "RCGPU_UPLOAD_THREADS", "RCGPU_DEC_STATES_SLACK"}      # sizing and tracing: same bytes
    for path in (api.LIB_PATH, os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")):
        names = set(re.findall(rb"RCGPU_[A-Z0-9_]+", open(path, "rb").read()))
        names = {n.decode() for n in names if not n.startswith((b"RCGPU_PIX_", b"RCGPU_FLAG_", b"RCGPU_RC_WHOLE", b"RCGPU_KEPT_"))}
        assert names <= allowed, names - allowed
    for fn in os.listdir(csrc):
        if fn.endswith((".cpp", ".hip", ".h")):
            text = open(os.path.join(csrc, fn)).read()
            assert "#ifdef RCGPU_EXP" not in text, fn
            for m in re.finditer(r'[^_A-Z]getenv\("(RCGPU_[A-Z0-9_]+)"\)', text):
                assert m.group(1) in allowed, (fn, m.group(1))


@pytest.mark.skipif(api.lib().rcgpu_device_count() > 0, reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback(built, tmp_path):
    with pytest.raises(api.RcgpuError, match="no HIP device"):
        api.Ffv1Encoder(64, 48, synth.PIX_RGB16_BE, 384, 2, 2)
    with pytest.raises(api.RcgpuError, match="no HIP device"):
        api.FlacEncoder(2, 48000, 16)
    p = tmp_path / "a.dpx"
    p.write_bytes(synth.dpx_file(synth.components(16, 16, 3, 16), synth.PIX_RGB16_BE))
    out = api.Output(Streams=[api.Stream(FileName=str(p), Slices="4")])
    assert out.Process(str(tmp_path / "o.mkv")) != 0 and "no HIP device" in api.last_error()
    assert not (tmp_path / "o.mkv").exists()


def test_slices_golden_table(built):
    """Project/GNU/CLI/test/slices.sh:65-90 on its own 4624-row table, without FFmpeg."""
    valid = set(int(x) for x in open(os.path.join(G, "valid_slices.txt")).read().split())
    # pix_fmt -> (bitdepth, pixels per block of the DPX flavor FFmpeg writes for it; DPX.cpp:184-231)
    fmt = {"rgb24": (8, 1), "rgba": (8, 1), "rgb48be": (16, 1), "rgba64be": (16, 1), "gbrp10le": (10, 1), "gbrp12le": (12, 1),
           "gbrap10le": (16, 1),      # FFmpeg's DPX encoder has no 10-bit RGBA: the test file comes out deeper than 10 bit (table rows = 12^2)
           "gbrap12le": (12, 1)}
    L = api.lib()
    rows = 0
    for line in open(os.path.join(G, "slices.txt")):
        w, h, f, c = line.strip().split(":")
        w, h, c = int(w), int(h), int(c)
        bits, ppb = fmt[f]
        n = L.rcgpu_reference_slices(w, h, bits, ppb)
        assert c <= n <= c + c // 2, (line, n)
        assert n in valid, (line, n)
        nh, nv = api.slices_to_grid(n)
        assert nh * nv == n and nv <= nh < 2 * nv
        rows += 1
    assert rows == 4624
    # the factorisation accepts exactly the reference's list
    L.rcgpu_slices_to_grid.restype = C.c_int
    a, b = C.c_uint32(), C.c_uint32()
    ok = {n for n in range(2, 1025) if L.rcgpu_slices_to_grid(n, C.byref(a), C.byref(b)) == 0}
    assert ok == {v for v in valid if v <= 1024}


@pytest.mark.parametrize("v", [v for v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))["ffv1"]
                               if v["name"].startswith(("dpx", "tiff"))], ids=lambda v: v["name"])
def test_dpx_probe_matches_what_the_reference_printed(built, v):
    """vectors.json holds, per flavor, the flavor string, the -slices value and the vflip decision the REAL reference printed
    (tests/golden/make_golden.py); the probe must re-derive them from the header alone."""
    bits, nc, _, _ = synth.PIX_INFO[v["pixfmt"]]
    comp = synth.components(v["width"], v["height"], nc, bits, "film", seed=1)
    tiff = v["name"].startswith("tiff")
    d = synth.tiff_file(comp, v["pixfmt"]) if tiff else synth.dpx_file(comp, v["pixfmt"], flags=v["flags"])
    i = api.tiff_probe(d) if tiff else api.dpx_probe(d)
    assert (i.width, i.height, i.pixfmt, i.flags, i.line_bytes, i.slices, i.flavor.decode()) == \
           (v["width"], v["height"], v["pixfmt"], v["flags"], v["line_bytes"], v["slices"], v["flavor"])
    assert i.data_offset + i.data_size == len(d)
    # and back: the flavor string the reference printed names this pixel layout (what a decode-side binding starts from)
    import ctypes
    pf = ctypes.c_uint32(99)
    assert api.lib().rcgpu_pixfmt_from_flavor(v["flavor"].encode(), ctypes.byref(pf)) == 0 and pf.value == v["pixfmt"], v["flavor"]


def test_wav_probe_rejects_an_all_zero_fmt_chunk(built):
    """A fmt chunk with channels = block_align = avg = 0 is coherent with itself (0 == 0); the data chunk must not divide by it
    (the reference rejects such files, WAV.cpp:125-221)."""
    import struct
    fmt = struct.pack("<HHIIHH", 1, 0, 48000, 0, 0, 0)
    wav = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + 16) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", 16) + bytes(16)
    with pytest.raises(api.RcgpuError):
        api.wav_probe(wav)
    fmt = struct.pack("<HHIIHH", 1, 2, 0, 0, 4, 16)          # sample rate 0
    wav = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + 16) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", 16) + bytes(16)
    with pytest.raises(api.RcgpuError):
        api.wav_probe(wav)


@pytest.mark.parametrize("v", json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))["ffv1"], ids=lambda v: v["name"])
def test_config_from_record(built, v):
    """CodecPrivate of every golden stream -> slice grid, ec, coder, context model (parameters::Parse, FFV1_Parameters.cpp:23-183)."""
    rec = bytes.fromhex(v["config_record"])
    cfg = api.config_from_record(rec, v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["flags"])
    assert (cfg.num_h_slices, cfg.num_v_slices, cfg.slicecrc, cfg.context, cfg.coder) == (v["num_h"], v["num_v"], 1, 1, v["coder"])
    bad = bytearray(rec); bad[3] ^= 0x10
    with pytest.raises(RuntimeError):
        api.config_from_record(bytes(bad), v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["flags"])     # CRC
    other = synth.PIX_Y8 if v["pixfmt"] != synth.PIX_Y8 else synth.PIX_RGB16_BE
    with pytest.raises(RuntimeError):
        api.config_from_record(rec, v["width"], v["height"], other, v["line_bytes"], 0)                         # stream does not match the files


def test_hostile_records_and_files_are_refused_not_trusted(built):
    """Mutated CodecPrivates re-sealed with a valid CRC, and mutated / truncated file headers: every call returns (an error or a
    sane configuration).  tools/asan_cpu.sh runs this under ASan + UBSan; tools/fuzz/ holds the long-running version."""
    import ctypes as C
    rng = np.random.default_rng(7)
    vs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))["ffv1"]
    for v in vs[:6]:
        rec = bytes.fromhex(v["config_record"])
        for _ in range(1500):
            b = bytearray(rec)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, len(b) - 4))] = int(rng.integers(0, 256))
            c = api.lib().rcgpu_crc32_ffv1(bytes(b[:-4]), len(b) - 4)
            b[-4:] = c.to_bytes(4, "big")
            try:
                cfg = api.config_from_record(bytes(b), v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["flags"])
                assert 1 <= cfg.num_h_slices <= v["width"] and 1 <= cfg.num_v_slices <= v["height"] and cfg.slicecrc in (0, 1) and cfg.coder in (1, 2)
            except RuntimeError:
                pass
    files = [synth.dpx_file(synth.components(33, 7, 3, 10, "film", seed=2), synth.PIX_RGB10_FILLEDA_BE),
             synth.tiff_file(synth.components(32, 16, 3, 16, "film", seed=1), synth.PIX_RGB16_LE, trailer=b"xx"),
             synth.exr_file(synth.components(32, 16, 3, 16, "film", seed=1)),
             synth.wav_file(rng.integers(-1000, 1000, size=(300, 2)).astype(np.int32), 16)]
    for d in files:
        for _ in range(1500):
            n = len(d) if rng.integers(0, 4) else int(rng.integers(0, min(len(d), 2048)))
            b = bytearray(d[:n])
            for _ in range(int(rng.integers(1, 6))):
                if n:
                    b[int(rng.integers(0, min(n, 2048)))] = int(rng.integers(0, 256))
            for probe in (api.dpx_probe, api.tiff_probe, api.exr_probe, api.wav_probe):
                try:
                    i = probe(bytes(b))
                    if hasattr(i, "data_offset"):
                        assert i.data_offset + i.data_size <= len(b)
                except RuntimeError:
                    pass


def test_a_line_that_does_not_fit_32_bits_is_refused(built):
    """width x bytes per pixel is a 64-bit product everywhere: a DPX header naming 2^29 RGBA16 pixels on one line (4 GiB a line: the 32-bit
    product is 0, and `offset + 0 <= size` holds) is refused by the probe, and by the encoder and the decoder before any device is looked for."""
    d = bytearray(synth.dpx_file(synth.components(8, 2, 4, 16, "film", seed=3), synth.PIX_RGBA16_BE))
    assert api.dpx_probe(bytes(d)).width == 8
    d[772:776] = (1 << 29).to_bytes(4, "big"); d[776:780] = (1).to_bytes(4, "big")
    with pytest.raises(RuntimeError, match="does not fit 32 bits"):
        api.dpx_probe(bytes(d))
    for make in (lambda: api.Ffv1Encoder(1 << 29, 1, synth.PIX_RGBA16_BE, 0, 1, 1), lambda: api.Ffv1Decoder(1 << 29, 1, synth.PIX_RGBA16_BE, 0, 1, 1, 1, 1)):
        with pytest.raises(RuntimeError, match="does not fit 32 bits|2\\^30 pixels"):
            make()


def test_exr_probe(built):
    v = [v for v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))["ffv1"] if v["name"].startswith("exr")][0]
    d = synth.exr_file(synth.components(v["width"], v["height"], 3, 16, "film", seed=1), trailer=b"tail")
    i = api.exr_probe(d)
    assert (i.width, i.height, i.pixfmt, i.line_bytes, i.slices, i.flavor.decode(), i.framerate) == \
           (v["width"], v["height"], synth.PIX_EXR_RGB16, v["line_bytes"], v["slices"], v["flavor"], 24.0)      # flavor / slices as the reference printed them
    assert d[i.data_offset + i.data_size:] == b"tail"
    # what exr::ParseBuffer refuses (EXR.cpp:199-566) is refused here too, never a crash
    for bad in (d[:100], d[:4] + b"\x02\x02\x00\x00" + d[8:], d.replace(b"compression\0compression\0\x01\0\0\0\0", b"compression\0compression\0\x01\0\0\0\x03"),
                d.replace(b"lineOrder\0lineOrder\0\x01\0\0\0\0", b"lineOrder\0lineOrder\0\x01\0\0\0\x01"), d.replace(b"pixelAspectRatio", b"pixelAspectRatiX"), d[:-30000]):
        with pytest.raises(RuntimeError):
            api.exr_probe(bad)


def test_probes(built):
    comp = synth.components(4096 // 16, 2160 // 16, 3, 16, "film", seed=3)
    d = synth.dpx_file(comp, synth.PIX_RGB16_BE, fps=25.0)
    i = api.dpx_probe(d)
    assert (i.width, i.height, i.pixfmt, i.data_offset, i.line_bytes, i.framerate) == (256, 135, synth.PIX_RGB16_BE, 2048, 1536, 25.0)
    assert i.flavor == b"DPX/Raw/RGB/16bit/U/BE" and i.data_size == 1536 * 135
    le = synth.dpx_file(synth.components(50, 38, 3, 10), synth.PIX_RGB10_FILLEDA_LE)
    assert api.dpx_probe(le).flavor == b"DPX/Raw/RGB/10bit/U/LE/FilledA"                  # DPX.cpp:762-778
    assert api.dpx_probe(synth.dpx_file(synth.components(16, 16, 3, 8), synth.PIX_RGB8)).flavor == b"DPX/Raw/RGB/8bit/U/LE"
    assert api.lib().rcgpu_reference_slices(4096, 2160, 16, 1) == 576 and api.lib().rcgpu_reference_slices(2048, 1556, 10, 1) == 64
    t = synth.tiff_file(synth.components(40, 30, 3, 16), synth.PIX_RGB16_LE, trailer=b"1234567")
    ti = api.tiff_probe(t)
    assert (ti.width, ti.height, ti.line_bytes, ti.flavor) == (40, 30, 240, b"TIFF/Raw/RGB/16bit/U/LE")
    assert t[ti.data_offset + ti.data_size:] == b"1234567"
    w = synth.wav_file(synth.pcm_samples(1000, 6, 24), 24, 48000, extensible=True)
    wi = api.wav_probe(w)
    assert (wi.channels, wi.sample_rate, wi.bits_per_sample, wi.data_size, wi.flavor) == (6, 48000, 24, 18000, b"WAV/PCM/48kHz/24bit/6ch/S/LE")
    # malformed input is an error, never a crash
    for bad in (b"", b"SDPX", d[:1000], b"XXXX" + d[4:], d[:3000], t[:100], b"RIFF\x00\x00\x00\x00WAVE", w[:60]):
        for probe in (api.dpx_probe, api.tiff_probe, api.wav_probe):
            try:
                probe(bad)
            except api.RcgpuError:
                pass


def test_hash_known_answers(built):
    for msg in (b"", b"a", b"abc", b"x" * 55, b"y" * 56, b"z" * 64, bytes(range(256)) * 37):
        assert api.md5(msg) == hashlib.md5(msg).digest()
    L = api.lib()
    assert L.rcgpu_crc32_ffv1(b"123456789", 9) == 0x89A1897F          # CRC-32/MPEG-2 with init 0: poly 04C11DB7, no reflection, no xorout
    data = os.urandom(1000)
    crc = L.rcgpu_crc32_ffv1(data, len(data))
    assert L.rcgpu_crc32_ffv1(data + struct.pack(">I", crc), len(data) + 4) == 0   # parity property FFV1 relies on


def _ebml(buf, pos, end):
    """Minimal EBML walker -> list of (id, data_start, data_end)."""
    out = []
    while pos < end:
        b = buf[pos]
        n = 8 - b.bit_length() + 1
        eid = int.from_bytes(buf[pos:pos + n], "big")
        pos += n
        b = buf[pos]
        n = 8 - b.bit_length() + 1
        size = int.from_bytes(buf[pos:pos + n], "big") & ((1 << (7 * n)) - 1)
        pos += n
        out.append((eid, pos, pos + size))
        pos += size
    return out


def test_muxer_layout(built, tmp_path):
    """What the reference's reader needs (Matroska.cpp:863-873, :934-953, :1007-1030, :1259-1277)."""
    path = str(tmp_path / "m.mkv")
    mux = api.MkvMuxer(path)
    tv = mux.add_video(b"\x01\x02\x03", 4096, 2160, 24000, 1001)
    ta = mux.add_audio(b"fLaC" + bytes(38), 6, 48000, 24)
    mux.add_attachment("side.txt", b"hello")
    mux.add_attachment("RAWcooked reversibility data", b"\x1a\x45\xdf\xa3rest")
    mux.add_tag(tv, "WARNING", "Pixel content is IEEE 754 floating-point format")          # Output.cpp:123
    with pytest.raises(api.RcgpuError):
        mux.add_tag(7, "A", "B")
    mux.begin()
    big = os.urandom(3 << 20)
    for i in range(3):
        mux.write_block(ta, i * 96000000, b"A" * 100)
        mux.write_block(tv, i * 1001 * 10 ** 9 // 24000, big if i == 1 else b"V" * 500)
    mux.close()
    buf = open(path, "rb").read()
    top = _ebml(buf, 0, len(buf))
    assert [e[0] for e in top] == [0x1A45DFA3, 0x18538067] and top[1][2] == len(buf)      # Segment size known and exact
    seg = _ebml(buf, top[1][1], top[1][2])
    ids = [e[0] for e in seg]
    assert ids.index(0x1654AE6B) < ids.index(0x1941A469) < ids.index(0x1F43B675)           # Tracks < Attachments < first Cluster
    assert ids[-1] == 0x1C53BB6B                                                           # Cues last
    tag = _ebml(buf, *_ebml(buf, *[e for e in seg if e[0] == 0x1254C367][0][1:])[0][1:])
    targets = {e[0]: buf[e[1]:e[2]] for e in _ebml(buf, *[e for e in tag if e[0] == 0x63C0][0][1:])}
    simple = {e[0]: buf[e[1]:e[2]] for e in _ebml(buf, *[e for e in tag if e[0] == 0x67C8][0][1:])}
    assert simple[0x45A3] == b"WARNING" and simple[0x4487].startswith(b"Pixel content is IEEE 754")
    seek = [{x[0]: buf[x[1]:x[2]] for x in _ebml(buf, s_[1], s_[2])} for s_ in _ebml(buf, *[e for e in seg if e[0] == 0x114D9B74][0][1:])]
    for sk in seek:                                                                        # every SeekHead entry points at the element it names
        at = top[1][1] + int.from_bytes(sk[0x53AC], "big")
        assert buf[at:at + len(sk[0x53AB])] == sk[0x53AB]
    assert {sk[0x53AB] for sk in seek} >= {bytes.fromhex("1254C367"), bytes.fromhex("1654AE6B"), bytes.fromhex("1C53BB6B")}
    tracks = _ebml(buf, *[e for e in seg if e[0] == 0x1654AE6B][0][1:])
    assert len(tracks) == 2
    ent = {e[0]: buf[e[1]:e[2]] for e in _ebml(buf, tracks[0][1], tracks[0][2])}
    assert targets[0x63C5] == ent[0x73C5]                                                  # the tag targets the video track's UID
    assert ent[0x86] == b"V_FFV1" and ent[0x63A2] == b"\x01\x02\x03" and ent[0xD7] == b"\x01"
    vid = {e[0]: buf[e[1]:e[2]] for e in _ebml(buf, *[e for e in _ebml(buf, tracks[0][1], tracks[0][2]) if e[0] == 0xE0][0][1:])}
    assert vid[0xB0] == b"\x10\x00" and vid[0xBA] == b"\x08\x70"                           # 2-byte uints
    blocks = []
    for e in seg:
        if e[0] == 0x1F43B675:
            for c in _ebml(buf, e[1], e[2]):
                assert c[0] in (0xE7, 0xA3)                                                # Timestamp + SimpleBlocks only
                if c[0] == 0xA3:
                    blocks.append((buf[c[1]] & 0x7F, c[2] - c[1] - 4))
    assert blocks == [(2, 100), (1, 500), (2, 100), (1, 3 << 20), (2, 100), (1, 500)]
    att = _ebml(buf, *[e for e in seg if e[0] == 0x1941A469][0][1:])
    datas = [{x[0]: buf[x[1]:x[2]] for x in _ebml(buf, a[1], a[2])} for a in att]
    assert datas[1][0x465C].startswith(b"\x1a\x45\xdf\xa3") and datas[0][0x466E] == b"side.txt"
    with pytest.raises(api.RcgpuError):
        api.MkvMuxer(path, overwrite=False)


def test_argv_front_end_rejects_what_it_cannot_do(built, tmp_path):
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    r = subprocess.run([shim, "-version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("rcgpu version")                       # Main.cpp:751-774 parses "<name> version <x>"
    p = tmp_path / "a.dpx"
    p.write_bytes(synth.dpx_file(synth.components(16, 16, 3, 16), synth.PIX_RGB16_BE))
    for extra in (["-coder", "0"], ["-level", "0"], ["-level", "1", "-slices", "4"], ["-c:v", "ffv1_nvenc"], ["-g", "2"], ["-vf", "scale=8:8"], ["-vf", "hwupload,transpose"],
                  ["-init_hw_device", "cuda=cu:0"], ["-init_hw_device", "vulkan=vk:x"], ["-init_hw_device", "vulkanize"]):
        r = subprocess.run([shim, "-i", str(p), "-c:v", "ffv1", "-coder", "1", "-level", "3", "-g", "1"] + extra + ["-f", "matroska", str(tmp_path / "o.mkv")],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "Error: " in r.stderr                                  # helpers.sh:81 greps for "Error:"


def test_wav_probe_describes_what_only_copy_can_carry(built):
    s = synth.pcm_samples(500, 2, 24)
    wi = api.wav_probe(synth.wav_file(s, 32, 44100))
    assert (wi.bits_per_sample, wi.format_tag, wi.flavor) == (32, 1, b"WAV/PCM/44kHz/32bit/2ch/S/LE")      # WAV_Tested, WAV.cpp:142-145
    wf = api.wav_probe(synth.wav_file(s, 32, 48000, float32=True, extensible=True))
    assert (wf.bits_per_sample, wf.format_tag, wf.flavor) == (32, 3, b"WAV/PCM/48kHz/32bit/2ch/F/LE")
    assert api.wav_probe(synth.wav_file(s, 24)).format_tag == 1


def _ebml_find(buf, wanted):
    """payloads of every element with id `wanted` (depth-first through the master elements we write)"""
    masters = {0x18538067, 0x1654AE6B, 0xAE, 0xE1, 0x1F43B675}
    out, stack = [], [(0, len(buf))]
    while stack:
        pos, end = stack.pop()
        while pos < end:
            b = buf[pos]; n = 1 + (8 - b.bit_length()); eid = int.from_bytes(buf[pos:pos + n], "big"); pos += n
            b = buf[pos]; n = 1 + (8 - b.bit_length()); size = int.from_bytes(buf[pos:pos + n], "big") & ((1 << (7 * n)) - 1); pos += n
            if eid == wanted:
                out.append((pos, bytes(buf[pos:pos + size])))
            if eid in masters:
                stack.append((pos, pos + size))
            pos += size
    return [p for _, p in sorted(out)]


@pytest.mark.parametrize("kind", ["s16", "s32", "f32"])
def test_audio_copy_needs_no_device_and_keeps_every_byte(built, tmp_path, kind):
    """`-c:a copy` (test/pcm.sh; forced by the reference above 24 bits, CLI/Main.cpp:300-317): a PCM track whose blocks are the
    WAV's data chunk cut at whole sample frames.  Nothing is coded, so this is the one job that runs without a GPU."""
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    s = synth.pcm_samples(30011, 6 if kind == "f32" else 2, 16 if kind == "s16" else 24)
    wav = synth.wav_file(s, 16, 48000) if kind == "s16" else synth.wav_file(s, 32, 48000, float32=(kind == "f32"))
    info = api.wav_probe(wav)
    (tmp_path / "a.wav").write_bytes(wav)
    out = tmp_path / "o.mkv"
    r = subprocess.run([shim, "-xerror", "-i", str(tmp_path / "a.wav"), "-c:a", "copy", "-f", "matroska", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    mkv = out.read_bytes()
    assert _ebml_find(mkv, 0x86) == [b"A_PCM/FLOAT/IEEE" if kind == "f32" else b"A_PCM/INT/LIT"]
    assert _ebml_find(mkv, 0x63A2) == []                                                    # no CodecPrivate
    assert [int.from_bytes(x, "big") for x in _ebml_find(mkv, 0x9F) + _ebml_find(mkv, 0x6264)] == [info.channels, info.bits_per_sample]
    blocks = _ebml_find(mkv, 0xA3)
    assert all(b[0] == 0x81 and (len(b) - 4) % info.block_align == 0 and len(b) - 4 <= 4096 for b in blocks)
    assert b"".join(b[4:] for b in blocks) == wav[info.data_offset:info.data_offset + info.data_size]
    if kind != "s16":                                                                       # FLAC cannot carry these: CLI/Main.cpp:308-314
        r = subprocess.run([shim, "-i", str(tmp_path / "a.wav"), "-c:a", "flac", "-y", "-f", "matroska", str(out)], capture_output=True, text=True)
        assert r.returncode != 0 and "Error: " in r.stderr                                   # here: no device; on a GPU box: "use -c:a copy"


@pytest.mark.parametrize("kind", ["s16-by-hand", "s32", "f32"])
def test_reference_round_trips_copied_audio(built, refbin, tmp_path, kind):
    """The reference drives the shim (`--bin-name`) and checks the result itself; with 32-bit input it adds `-c:a copy` on its own."""
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    os.makedirs(tmp_path / "pkg")
    s = synth.pcm_samples(20001, 2, 16 if kind == "s16-by-hand" else 24)
    (tmp_path / "pkg" / "1.wav").write_bytes(synth.wav_file(s, 16) if kind == "s16-by-hand" else synth.wav_file(s, 32, float32=(kind == "f32")))
    cmd = [refbin, "--bin-name", shim] + (["-c:a", "copy"] if kind == "s16-by-hand" else []) + ["-y", "--check", "pkg"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
    assert r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout, r.stdout + r.stderr


def test_linked_reference_reports_the_missing_device(built, tmp_path):
    """Route B compiled (oracle/_ref/rawcooked_linked): without a GPU the reference's own error path shows rcgpu_encode's message and
    exit status -- there is no CPU encode path to fall back to."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/rawcooked_linked not built (needs /root/reference)")
    if api.lib().rcgpu_device_count() > 0:
        pytest.skip("a HIP device is visible")
    os.makedirs(tmp_path / "pkg" / "img")
    for i in range(2):
        (tmp_path / "pkg" / "img" / ("f_%06d.dpx" % i)).write_bytes(synth.dpx_file(synth.components(64, 48, 3, 16, "film", seed=i), synth.PIX_RGB16_BE, frame_index=i))
    r = subprocess.run([exe, "--hash", "-y", "pkg"], cwd=tmp_path, capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
    assert r.returncode != 0 and "Error: no HIP device available -- rcgpu has no CPU encode path" in r.stderr, r.stdout + r.stderr
    assert not os.path.exists(tmp_path / "pkg.mkv")


@pytest.mark.parametrize("context,coder,pixfmt", [(0, 1, synth.PIX_RGB16_BE), (1, 1, synth.PIX_RGB16_BE), (2, 1, synth.PIX_RGB10_FILLEDA_BE), (0, 2, synth.PIX_RGBA16_LE), (1, 2, synth.PIX_Y16_BE)])
def test_config_from_stream_reads_the_table_set_from_the_first_slice_header(built, context, coder, pixfmt):
    """What the record cannot say -- which quantisation table set the planes use (FFV1_Slice.cpp:159-168) -- comes from the first packet."""
    import ctypes
    import oracle_binding as ob
    w, h = 96, 64
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    p = ob.Params(w, h, pixfmt, 3, 2, 1, context, 0, coder)
    pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, "film", seed=4), pixfmt, True)
    rec, pk = ob.config_record(p), ob.encode_payload(p, pl, line_bytes)
    for hint in (0, 1):
        cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 0, 0, 0, hint, 1, 0, 0, 0, 0, 0)
        assert api.lib().rcgpu_ffv1_config_from_stream(rec, len(rec), pk, len(pk), ctypes.byref(cfg)) == 0, api.last_error()
        assert (cfg.context, cfg.coder, cfg.num_h_slices, cfg.num_v_slices, cfg.slicecrc) == (context, coder, 3, 2, 1)
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0)
    assert api.lib().rcgpu_ffv1_config_from_stream(rec, len(rec), pk[:4], 4, ctypes.byref(cfg)) != 0


_VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))
_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ext_record(v):
    return open(os.path.join(_G, v["config_record_file"]), "rb").read() if "config_record_file" in v else bytes.fromhex(v.get("config_record", ""))


@pytest.mark.parametrize("v", _VEC["ffv1"] + _VEC["ffv1_ext"], ids=lambda v: v["name"])
def test_stream_parse_reads_what_the_reference_reads(built, v):
    """rcgpu_ffv1_stream_parse = parameters::Parse (FFV1_Parameters.cpp:23-183, tables :206-253) + the first slice header's
    quant_table_set_index values (FFV1_Slice.cpp:158-168), for every golden stream the real reference checked: the defaults FFmpeg writes,
    and 1-8 table sets of arbitrary tables, a set per plane group, transmitted transitions, coded initial states, version 0 / 1 headers."""
    import oracle_binding as ob
    import ctypes
    rec = _ext_record(v)
    pk = open(os.path.join(_G, v["frames"][0]["packet"]), "rb").read()
    s = api.Ffv1Stream(rec, pk)
    i = s.info()
    ext = v.get("ext")
    bits, nc, _, _ = synth.PIX_INFO[v["pixfmt"]]
    assert (i.colorspace_type, i.bits_per_raw_sample, i.chroma_planes, i.alpha_plane) == (int(nc != 1), bits, int(nc != 1), int(nc == 4))
    assert i.quant_table_set_index_count == (nc - 1 if nc != 1 else 2)
    if ext is None:
        assert (i.version, i.micro_version, i.coder_type, i.num_h_slices, i.num_v_slices, i.quant_table_set_count, i.ec, i.intra) == (3, 4, v["coder"], v["num_h"], v["num_v"], 2, 1, 1)
        assert list(i.quant_table_set_index)[:i.quant_table_set_index_count] == [1] * i.quant_table_set_index_count and list(i.states_coded) == [0] * 8
        assert list(i.context_count)[:2] == ([666, 7563] if bits <= 8 else [365, 5063])
    else:
        e = ob.stream_ext_from_vector(v, _G)
        version = ext.get("version", 3)
        assert (i.version, i.coder_type, i.quant_table_set_count, i.ec) == (version, 2 if "one_state" in ext else 1, len(ext["sets"]), v["ec"])
        assert (i.num_h_slices, i.num_v_slices) == (v["num_h"], v["num_v"])
        assert i.intra == (ext.get("intra", 1) if version == 3 else 0)
        want_idx = list(ext.get("set_index", (0, 0, 0))) if version == 3 else [0, 0, 0]
        assert list(i.quant_table_set_index)[:i.quant_table_set_index_count] == want_idx[:i.quant_table_set_index_count]
        assert list(i.context_count)[:len(ext["sets"])] == [ob.lib().ffv1o_ext_context_count(ctypes.byref(e), k) for k in range(len(ext["sets"]))]
        assert [k for k in range(8) if i.states_coded[k]] == sorted(int(k) for k in ext.get("initial_states", {}))
    s.close()
    if rec:
        bad = bytearray(rec); bad[len(rec) // 2] ^= 0x10
        with pytest.raises(api.RcgpuError, match="CRC"):
            api.Ffv1Stream(bytes(bad), pk)
        with pytest.raises(api.RcgpuError):
            api.Ffv1Stream(rec, b"")                                                     # the first packet is part of the description
    else:
        with pytest.raises(api.RcgpuError):
            api.Ffv1Stream(b"", pk[:3])


def test_stream_parse_survives_hostile_input(built):
    """Mutated records (re-sealed with a valid CRC) and mutated first packets, incl. the version 0 / 1 headers inside the packet and records
    with a million coded initial states cut short: an error or a description whose numbers are in range.  (tools/fuzz/ runs the long version.)"""
    rng = np.random.default_rng(11)
    for v in _VEC["ffv1_ext"]:
        rec = _ext_record(v)
        pk = open(os.path.join(_G, v["frames"][0]["packet"]), "rb").read()
        for k in range(120 if len(rec) > 4096 else 600):
            r, q = bytearray(rec), bytearray(pk[:256])
            tgt = r if (rec and k % 3) else q
            for _ in range(int(rng.integers(1, 5))):
                tgt[int(rng.integers(0, max(1, len(tgt) - 4)))] = int(rng.integers(0, 256))
            if rec and rng.integers(0, 4) == 0:
                r = r[:int(rng.integers(5, len(r)))]
            if rec:
                r[-4:] = api.lib().rcgpu_crc32_ffv1(bytes(r[:-4]), len(r) - 4).to_bytes(4, "big")
            try:
                s = api.Ffv1Stream(bytes(r), bytes(q))
            except api.RcgpuError:
                continue
            i = s.info()
            assert i.version in (0, 1, 3) and 1 <= i.quant_table_set_count <= 8 and i.ec <= 1 and i.intra <= 1 and i.coder_type in (1, 2)
            assert all(1 <= c <= 16384 for c in list(i.context_count)[:i.quant_table_set_count])
            assert all(x < i.quant_table_set_count for x in list(i.quant_table_set_index)[:i.quant_table_set_index_count])
            s.close()


from range_writer import RangeEncoder as _RangeEncoder


def _record_with_run(run_minus1: int, tail=b"") -> bytes:
    """A version 3 record (RGB, 16 bit, 1 x 1 slices, one table set) whose first table starts with one entry and goes on with `run_minus1`."""
    e, st = _RangeEncoder(), [128] * 32
    for v in (3, 4, 1, 1, 16):                      # version, micro_version, coder_type, colorspace_type, bits_per_raw_sample
        e.u(st, v)
    e.bit(st, 0, 1); e.u(st, 0); e.u(st, 0); e.bit(st, 0, 0)        # chroma_planes, subsampling, alpha_plane
    e.u(st, 0); e.u(st, 0); e.u(st, 1)              # 1 x 1 slices, one quantisation table set
    q = [128] * 32
    e.u(q, 0); e.u(q, run_minus1)
    body = e.done() + tail
    return body + api.lib().rcgpu_crc32_ffv1(body, len(body)).to_bytes(4, "big")


def test_stream_parse_does_its_sums_in_the_reference_s_width(built):
    """`k + len_minus1 >= 128` is a size_t sum in parameters::QuantizationTable (FFV1_Parameters.cpp:231): a run of 2^32 - 1 after the
    first entry must not wrap to 0 and then be written; and an exponent beyond 31 condemns the stream (ForceUnderrun, FFV1_RangeCoder.cpp:
    114-118) rather than reading as 0."""
    from rfc9043_validator import RangeDecoder, default_state_transition
    rec = _record_with_run(0xFFFFFFFF, b"\0" * 64)
    rd, st, q = RangeDecoder(rec[:-4], default_state_transition()), [128] * 32, [128] * 32          # the writer above against the validator's reader
    assert [rd.symbol(st, False) for _ in range(5)] == [3, 4, 1, 1, 16] and rd.bit(st, 0) == 1
    assert [rd.symbol(st, False) for _ in range(2)] == [0, 0] and rd.bit(st, 0) == 0 and [rd.symbol(st, False) for _ in range(3)] == [0, 0, 1]
    assert [rd.symbol(q, False) for _ in range(2)] == [0, 0xFFFFFFFF]
    for run in (126, 0xFFFFFF80, 5):
        try:
            api.Ffv1Stream(_record_with_run(run, b"\0" * 64), bytes(16)).close()               # parsed or refused; the table after it is zeros' doing
        except api.RcgpuError:
            pass
    with pytest.raises(api.RcgpuError, match="bad quantisation table"):
        api.Ffv1Stream(_record_with_run(0xFFFFFFFF, b"\0" * 64), bytes(16))
    with pytest.raises(api.RcgpuError):
        api.Ffv1Stream(_record_with_run(1 << 32, b"\0" * 64), bytes(16))


def test_probes_agree_with_the_reference_s_parsers(built):
    """3601 small files -- DPX of all 22 flavors, TIFF of eight, EXR, WAV of eight (32-bit integer and float among them), and seeded mutations of their headers (tests/golden/make_probe_golden.py makes them
    again here) -- with what the REAL reference's wav, dpx, tiff and exr parsers said of each (oracle/ref_probe.cpp drives them in the CLI's
    order; tests/golden/probe_cases.txt): whatever the reference would hand to its encoder the shim's probes take, from the same parser, as
    the same flavor, with the same -slices.  (The other way round is not asked for: the shim is only ever called for what the reference
    supports.)  One class apart: a TIFF whose Compression tag is not 1 -- the reference only tests that the tag is there (TIFF.cpp:465,557),
    FFmpeg would try to decompress; the probe refuses -- as it refuses a TIFF whose strips are not the picture's size or whose line does not
    fit 32 bits, which the reference's 32-bit sums let through on hostile headers."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_probe_golden", os.path.join(_G, "make_probe_golden.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cases = mod.cases()
    lines = open(os.path.join(_G, "probe_cases.txt")).read().splitlines()
    blob = b"".join(struct.pack("<I", len(c)) + c for c in cases)
    assert lines[0] == "sha256 " + hashlib.sha256(blob).hexdigest(), "the generator no longer makes the files the reference was shown: run tests/golden/make_probe_golden.py"
    lines = lines[1:]
    assert len(lines) == len(cases) > 3500
    probes = {"wav": api.wav_probe, "dpx": api.dpx_probe, "tiff": api.tiff_probe, "exr": api.exr_probe}
    supported = stricter = 0
    for n, (data, line) in enumerate(zip(cases, lines)):
        w = line.split()
        if w[1] != "supported=1":
            continue
        supported += 1
        try:
            i = probes[w[0]](data)
        except RuntimeError as ex:
            # (what FFmpeg, whose place the shim takes, would not have encoded either)
            assert w[0] == "tiff" and any(t in str(ex) for t in ("compressed content", "strip sizes do not match", "does not fit 32 bits")), (n, line, str(ex))
            stricter += 1
            continue
        assert i.flavor.decode() == w[2][len("flavor="):] and getattr(i, "slices", 0) == int(w[3][len("slices="):]), (n, line, i.flavor, getattr(i, "slices", 0))
        for other in ("wav", "dpx", "tiff", "exr"):                         # and no earlier parser of the CLI's order claims the file
            if other == w[0]:
                break
            with pytest.raises(RuntimeError):
                probes[other](data)
    assert supported > 2400 and stricter <= 6


_FNV0 = 0xcbf29ce484222325


def _fnv(h, data):
    for x in data:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def _parse_cases():
    blob = open(os.path.join(_G, "parse_cases.bin"), "rb").read()
    lines = open(os.path.join(_G, "parse_cases.txt")).read().splitlines()
    cases, o = [], 0
    while o < len(blob):
        a, b = struct.unpack_from("<II", blob, o)
        cases.append((blob[o + 8:o + 8 + a], blob[o + 8 + a:o + 8 + a + b]))
        o += 8 + a + b
    assert len(cases) == len(lines)
    return cases, lines


def _reference_said(line):
    """A line of oracle/ref_ffv1_parse.cpp: None when the reference refused, else (fields, transitions hash, [(tables hash, contexts, states hash)],
    underrun, (key, header ok, indexes, underrun, message) or None)."""
    w = line.split()
    if w[0] == "ERR":
        return None
    f = [int(x) for x in w[1:18]]     # version micro coder colorspace bits chroma log2h log2v alpha num_h num_v sets ec intra index_count custom underrun
    rest, sets, k = w[18:], [], 1
    while k < len(rest) and rest[k][0] == "Q":
        sets.append((int(rest[k][1:], 16), int(rest[k + 1][1:]), int(rest[k + 2][1:], 16)))
        k += 3
    hdr = None
    if k < len(rest):
        hdr = (int(rest[k][1:]), int(rest[k + 1][1:]), [int(x) for x in rest[k + 2][1:].split(",") if x], int(rest[k + 3][1:]), rest[k + 4][1:])
    return [f[0], f[1], f[2], f[3], f[4], f[5], f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15]], int(rest[0][1:], 16), sets, f[16], hdr


def _library_says(rec, pk):
    s = api.Ffv1Stream(rec, pk)
    i = s.info()
    fields = [i.version, i.micro_version, 1, i.colorspace_type, i.bits_per_raw_sample, i.chroma_planes, i.alpha_plane, i.num_h_slices, i.num_v_slices,
              i.quant_table_set_count, i.ec, i.intra, i.quant_table_set_index_count, int(i.coder_type == 2)]
    t, sets = None, []
    for k in range(i.quant_table_set_count):
        one, q, init = s.tables(k)
        t = _fnv(_FNV0, one[1:])
        sets.append((_fnv(_FNV0, q.astype("<i4").tobytes()), i.context_count[k], _fnv(_FNV0, init if init else bytes([128]) * (i.context_count[k] * 32))))
    idx = list(i.quant_table_set_index)[:i.quant_table_set_index_count]
    s.close()
    return fields, t, sets, idx


def _hold_to_the_reference(cases, lines):
    both = 0
    for n, ((rec, pk), line) in enumerate(zip(cases, lines)):
        ref = _reference_said(line)
        try:
            mine = _library_says(rec, pk)
        except api.RcgpuError as ex:
            if ref is None:
                continue
            fields, _, _, underrun, hdr = ref
            # the library may refuse what the reference reads, never the other way round -- and only for these reasons: the record ran out of
            # bytes (the reference reads zeros on), a stream the device does not decode, or a first frame that is no key frame / whose first slice
            # header the reference itself flags / that ends inside it / that is shorter than any frame
            assert (ex.code == 3 and underrun) or ex.code == 20 or (ex.code == 8 and hdr and (not hdr[0] or not hdr[1] or hdr[3] or len(pk) < 8)), (n, str(ex), line)
            continue
        assert ref is not None, f"case {n}: the reference refuses ({line}) what the library reads"
        fields, t, sets, underrun, hdr = ref
        assert mine[0] == fields and mine[1] == t and mine[2] == sets, (n, mine, line)
        if hdr is not None:
            assert hdr[0] == 1 and hdr[1] == 1 and mine[3] == hdr[2], (n, mine[3], line)
        both += 1
    return both


def test_stream_parse_agrees_with_the_reference_s_reader(built):
    """3165 records and first frames -- the golden ones, ones written field by field on and beyond every limit (tests/range_writer.py), and seeded
    mutations of both -- with what the REAL reference's ffv1_frame::OutOfBand, parameters::Parse and slice::SliceHeader made of each
    (oracle/ref_ffv1_parse.cpp drives them; tests/golden/make_parse_golden.py): whatever the reference refuses the library refuses; whatever both
    read they read alike -- every field, every table value, every initial state, the table set of every plane group."""
    cases, lines = _parse_cases()
    assert len(cases) >= 3000
    assert _hold_to_the_reference(cases, lines) >= 500
    for v in _VEC["ffv1_ext"] + _VEC["ffv1"]:                                             # and the golden streams as they are: read by both
        rec = _ext_record(v) if "config_record_file" in v else bytes.fromhex(v.get("config_record", ""))
        if rec and len(rec) <= 700:
            k = cases.index((rec, open(os.path.join(_G, v["frames"][0]["packet"]), "rb").read()[:48]))
            assert _reference_said(lines[k]) is not None and _library_says(*cases[k])


def test_more_slice_rows_than_columns_stay_with_the_reference(built):
    """slice::SliceHeader compares slice_y with num_H_slices (FFV1_Slice.cpp:125): with 3 columns and 6 rows the reference reports
    FFV1-SLICE-slice_xywh for every slice from row 3 on (parse_cases.txt holds its verdict on such headers).  A device decoder that read them
    right would pass a file the reference fails: the stream is told apart (RCGPU_FFV1_UNSUPPORTED) and keeps the reference's decoder."""
    import range_writer as rw
    with pytest.raises(api.RcgpuUnsupported, match="more rows than columns"):
        api.Ffv1Stream(rw.record(h1=2, v1=5), rw.first_slice((0, 0)))
    api.Ffv1Stream(rw.record(h1=5, v1=2), rw.first_slice((0, 0))).close()
    api.Ffv1Stream(rw.record(h1=5, v1=5), rw.first_slice((0, 0))).close()


def test_stream_parse_agrees_with_the_reference_s_reader_on_fresh_mutations(built, tmp_path):
    """The same against the reference's reader itself (oracle/_ref/ref_ffv1_parse, where it was built), on 4000 mutations that are in no
    fixture.  RCGPU_FRESH_SEED=random draws a new seed every run (printed on failure; sixty thousand of them agreed in round 5); the suite's
    own run uses a fixed one, so that a gate is not decided by a draw."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_ffv1_parse")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_ffv1_parse not built (needs /root/reference)")
    import range_writer as rw
    seed = int.from_bytes(os.urandom(4), "little") if os.environ.get("RCGPU_FRESH_SEED") == "random" else int(os.environ.get("RCGPU_FRESH_SEED", "20261001"))
    rng = np.random.default_rng(seed)
    base, _ = _parse_cases()
    base = [c for c in base[:150] if len(c[0]) + len(c[1]) > 0]
    cases = []
    for _ in range(4000):
        rec, pk = base[int(rng.integers(0, len(base)))]
        r, q = bytearray(rec), bytearray(pk)
        for _ in range(int(rng.integers(1, 5))):
            tgt = r if (len(r) > 4 and rng.integers(0, 3)) else q
            if len(tgt):
                tgt[int(rng.integers(0, max(1, len(tgt) - (4 if tgt is r else 0))))] = int(rng.integers(0, 256))
        cases.append((rw.sealed(bytes(r[:-4])) if len(r) > 4 else bytes(r), bytes(q)))
    f = tmp_path / "cases.bin"
    f.write_bytes(b"".join(struct.pack("<II", len(r), len(p)) + r + p for r, p in cases))
    import ref_decode
    lines = subprocess.run([exe, str(f)], capture_output=True, text=True, check=True, env=ref_decode.clean_env()).stdout.splitlines()
    assert len(lines) == len(cases), f"seed {seed}"
    try:
        _hold_to_the_reference(cases, lines)
    except AssertionError as e:
        raise AssertionError(f"seed {seed}: {e}")


def test_decoder_for_stream_says_unsupported_before_it_looks_for_a_device(built):
    """What the reference decodes and the device does not is told apart from what is broken: RCGPU_FFV1_UNSUPPORTED (20), so that a binding
    leaves the stream to its own decoder (route C: ffv1_frame::Process stays on the slice pool).  intra = 0 is such a stream; a stream that
    does not describe the files' pixel format is an error; a fine stream on a box without a device fails for that reason only."""
    v = [x for x in _VEC["ffv1_ext"] if x["name"] == "ext_unsup_intra0_64x48"][0]
    s = api.Ffv1Stream(_ext_record(v), open(os.path.join(_G, v["frames"][0]["packet"]), "rb").read())
    assert s.info().intra == 0
    with pytest.raises(api.RcgpuUnsupported, match="intra = 0"):
        api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], max_batch=1, stream=s)
    with pytest.raises(api.RcgpuError, match="does not match the pixel format") as ei:
        api.Ffv1Decoder(v["width"], v["height"], synth.PIX_Y16_BE, v["width"] * 2, max_batch=1, stream=s)
    assert not isinstance(ei.value, api.RcgpuUnsupported)
    if api.lib().rcgpu_device_count() < 1:
        v = [x for x in _VEC["ffv1_ext"] if x["name"] == "ext_8sets_50x38"][0]
        s2 = api.Ffv1Stream(_ext_record(v), open(os.path.join(_G, v["frames"][0]["packet"]), "rb").read())
        with pytest.raises(api.RcgpuError, match="no HIP device"):
            api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], max_batch=1, stream=s2)


def test_sequence_structs_carry_their_size_and_one_input_callback(built):
    """rcgpu_sequence_io / rcgpu_sequence_options begin with struct_size: a caller compiled against another rcgpu.h is refused before any of its
    slots is called; read_frame and locate_frame are alternatives."""
    import ctypes as C
    cfg = api.Ffv1Config(64, 48, synth.PIX_RGB16_BE, 384, 2, 2, 1, 1, 2, 0, 0, 0, 1, 3)
    rf = api.READ_FRAME_FN(lambda user, frame, dst, n: 0)
    pd = api.PACKET_DONE_FN(lambda user, frame, data, n: 0)
    lf = api.LOCATE_FRAME_FN(lambda user, frame, n: 0)
    call = lambda io, opt: api.lib().rcgpu_ffv1_encode_sequence(C.byref(cfg), 4, C.byref(io), C.byref(opt) if opt is not None else None, None, None, None)
    good_opt = api.SequenceOptions(C.sizeof(api.SequenceOptions))
    assert call(api.SequenceIo(0, rf, api.PLACE_PACKET_FN(), pd, None), good_opt) == 1 and "another rcgpu.h" in api.last_error()
    assert call(api.SequenceIo(C.sizeof(api.SequenceIo) - 8, rf, api.PLACE_PACKET_FN(), pd, None), good_opt) == 1
    assert call(api.SequenceIo(C.sizeof(api.SequenceIo), rf, api.PLACE_PACKET_FN(), pd, None), api.SequenceOptions(4)) == 1 and "another rcgpu.h" in api.last_error()
    assert call(api.SequenceIo(C.sizeof(api.SequenceIo), rf, api.PLACE_PACKET_FN(), pd, None, lf), good_opt) == 1 and "alternatives" in api.last_error()
    if api.lib().rcgpu_device_count() < 1:                       # well-formed: gets as far as looking for a device
        assert call(api.SequenceIo(C.sizeof(api.SequenceIo), rf, api.PLACE_PACKET_FN(), pd, None), good_opt) == 4 and "no HIP device" in api.last_error()


def _reference_command_line(refbin, work, seqs, wavs=(), opts=()):
    """A package on disk and the ffmpeg command line the REAL reference writes for it (`-d`: display, do not run; CLI/Output.cpp:81-332)."""
    import shlex
    for name, (w, h, pixfmt, idx, kind) in seqs.items():
        os.makedirs(os.path.join(work, "pkg", name), exist_ok=True)
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        for i in idx:
            comp = synth.components(w, h, nc, bits, "film", seed=i % 5)
            data = synth.tiff_file(comp, pixfmt) if kind == "tif" else synth.exr_file(comp) if kind == "exr" else synth.dpx_file(comp, pixfmt, frame_index=i)
            open(os.path.join(work, "pkg", name, "f_%06d.%s" % (i, kind)), "wb").write(data)
    for wn, (ch, bits, rate, ns) in dict(wavs).items():
        open(os.path.join(work, "pkg", wn), "wb").write(synth.wav_file(np.random.default_rng(1).integers(-1000, 1000, size=(ns, ch)).astype(np.int32), bits, rate))
    from ref_decode import run_reference
    r = run_reference([refbin, "--no-check-padding", "--check", "-d", "-y"] + list(opts) + ["pkg"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    cmds = [ln for ln in r.stdout.splitlines() if ln.startswith("ffmpeg ")]
    assert len(cmds) == 1, r.stdout
    return shlex.split(cmds[0])


@pytest.mark.parametrize("name,seqs,wavs,opts,want", [
    ("plain", {"img": (64, 48, synth.PIX_RGB16_BE, range(6), "dpx")}, {}, [], ["video 6 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24/1 first pkg/img/f_000000.dpx"]),
    ("gapped: the reference's own file list", {"img": (64, 48, synth.PIX_RGB16_BE, [0, 1, 2, 5, 6, 9], "dpx")}, {}, [], ["video 6 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24/1 first "]),
    ("start number 86400", {"img": (64, 48, synth.PIX_RGB10_FILLEDA_BE, range(86400, 86405), "dpx")}, {}, [], ["video 5 frames 64x48 DPX/Raw/RGB/10bit/U/BE/FilledA slices 4x4 fps 24/1 first pkg/img/f_086400.dpx"]),
    ("tiff", {"img": (64, 48, synth.PIX_RGB16_LE, range(4), "tif")}, {}, [], ["video 4 frames 64x48 TIFF/Raw/RGB/16bit/U/LE slices 6x6 fps 24/1 first pkg/img/f_000000.tif"]),
    ("exr", {"img": (64, 48, synth.PIX_EXR_RGB16, range(4), "exr")}, {}, [], ["video 4 frames 64x48 EXR/Raw/RGB/16bit/F/BE slices 6x6 fps 24/1 first pkg/img/f_000000.exr"]),
    ("two sequences and two WAVs", {"a": (64, 48, synth.PIX_RGB16_BE, range(4), "dpx"), "b": (32, 16, synth.PIX_Y16_BE, range(8), "dpx")}, {"x.wav": (2, 16, 48000, 8000), "y.wav": (6, 24, 48000, 8000)}, [],
     ["video 4 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24/1 first pkg/a/f_000000.dpx", "video 8 frames 32x16 DPX/Raw/Y/16bit/U/BE slices 6x6 fps 24/1 first pkg/b/f_000000.dpx",
      "audio WAV/PCM/48kHz/16bit/2ch/S/LE 2 ch 48000 Hz 16 bit -> FLAC", "audio WAV/PCM/48kHz/24bit/6ch/S/LE 6 ch 48000 Hz 24 bit -> FLAC"]),
    ("-framerate 24000/1001", {"img": (64, 48, synth.PIX_RGB16_BE, range(6), "dpx")}, {}, ["-framerate", "24000/1001"], ["video 6 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24000/1001 first "]),
    ("-coder 2 -slices 16", {"img": (64, 48, synth.PIX_RGB16_BE, range(3), "dpx")}, {}, ["-coder", "2", "-slices", "16"], ["video 3 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 4x4 fps 24/1 first "]),
    ("-c:v ffv1_vulkan:1 = ffv1 on device 1", {"img": (64, 48, synth.PIX_RGB16_BE, range(3), "dpx")}, {}, ["-c:v", "ffv1_vulkan:1"],
     ["video 3 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24/1 first ", "device 1 (-init_hw_device)"]),
    ("-c:v ffv1_vulkan", {"img": (64, 48, synth.PIX_RGB16_BE, range(3), "dpx")}, {}, ["-c:v", "ffv1_vulkan"],
     ["video 3 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 6x6 fps 24/1 first ", "device 0 (-init_hw_device)"]),
    ("Y 10 bit", {"img": (50, 38, synth.PIX_Y10_FILLEDA_BE, range(3), "dpx")}, {}, [], ["video 3 frames 50x38 DPX/Raw/Y/10bit/U/BE/FilledA slices 4x4 fps 24/1 first "]),
    ("RGBA 12 bit packed", {"img": (48, 32, synth.PIX_RGBA12_PACKED_BE, range(3), "dpx")}, {}, [], ["video 3 frames 48x32 DPX/Raw/RGBA/12bit/U/BE/Packed slices 6x6 fps 24/1 first "]),
], ids=lambda x: x if isinstance(x, str) else "")
def test_shim_plans_what_the_reference_asks_for(built, refbin, tmp_path, name, seqs, wavs, opts, want):
    """The argv front end on the command lines the REAL reference writes (not on ones written from reading CLI/Output.cpp): the package is
    analysed by `rawcooked -d`, its ffmpeg command line -- quoting, option order, the file list it writes for a gapped sequence and all --
    goes to the shim with `-rcgpu_plan_only 1` (no device), and the plan is the package: every track, its frames, size, flavor, the slice
    grid of the reference's -slices, the frame rate."""
    work = str(tmp_path)
    argv = _reference_command_line(refbin, work, seqs, wavs, opts)
    if "ffv1_vulkan:1" in opts:          # the reference's GPU-selection surface as it writes it (CLI/Global.cpp:367-378, test/vulkan.sh:48-56)
        assert argv[argv.index("-init_hw_device") + 1] == "vulkan=vk:1" and argv[argv.index("-vf") + 1] == "hwupload" and argv[argv.index("-c:v", argv.index("-i")) + 1] == "ffv1_vulkan"
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    r = subprocess.run([shim] + argv[1:-1] + ["-rcgpu_plan_only", "1", argv[-1]], cwd=work, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    plan = [ln[len("rcgpu plan: "):] for ln in r.stdout.splitlines() if ln.startswith("rcgpu plan: ")]
    assert len(plan) == len(want) + 1 and plan[-1] == "output pkg.mkv, 0 attachment(s) + reversibility data", plan
    for got, w in zip(plan, want):
        assert got.startswith(w), (got, w)
    slices = int(argv[argv.index("-slices") + 1])
    nh, nv = api.slices_to_grid(slices)
    assert all(("slices %dx%d " % (nh, nv)) in ln for ln in plan if ln.startswith("video"))


def test_integration_patches_only_add_lines_and_hold_no_reference_source():
    """oracle/route_*.patch and linked_threadpool_h.patch are what a maintainer would add: every hunk adds lines and removes
    none (`diff -U0`: no context lines either, so no line of the reference's source is kept in this repository), and what the FFV1 / Matroska /
    file-writer hooks add stands under `#ifdef RCGPU_LINKED`."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = sorted(f for f in os.listdir(os.path.join(root, "oracle")) if f.endswith(".patch"))
    assert len(names) == 9
    for name in names:
        lines = open(os.path.join(root, "oracle", name), newline="").read().split("\n")
        body = [ln for ln in lines if ln and not ln.startswith(("--- ", "+++ ", "@@"))]
        assert body and all(ln.startswith(("+", "-")) for ln in body), name                    # no context lines
        assert not [ln for ln in body if ln.startswith("-")], name
        assert any("RCGPU_LINKED" in ln for ln in body), name


def test_traffic_json_describes_this_code():
    """profiles/traffic.json holds the HBM bytes the PMC passes measured -- and the sha256 of the kernel sources they ran on (tools/update_traffic.py).
    A kernel source that has changed since makes bench.py withhold the figure (`traffic: null, traffic_stale: true`); at the end of a round the
    passes are taken again and this holds."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert set(d["sources"]) == {"rawcooked_amd/csrc/ffv1_gpu.hip", "rawcooked_amd/csrc/ffv1_check.hip"}
    for name, want in d["sources"].items():
        if hashlib.sha256(open(os.path.join(root, name), "rb").read()).hexdigest() != want:
            # the source changed: fine while the KERNELS did not (host code inside a .hip file) -- the built library's device code is the one measured
            from rawcooked_amd import devcode
            assert d.get("device_code_sha256") and devcode.fatbin_sha256(os.path.join(root, "rawcooked_amd", "librcgpu.so")) == d["device_code_sha256"], \
                f"{name} and the kernels built from it changed since the PMC passes of {d.get('measured_on')}: bash tools/round.sh refresh, then bash tools/adopt_profiles.sh"
    for k in ("k_resolve", "k_rangecode", "k_dec_slices"):
        assert d[k]["per_frame_bytes"] > 0 and d[k]["source"] in ("ffv1_gpu.hip", "ffv1_check.hip")


def test_plan_only_analyses_a_job_without_a_device(built, tmp_path):
    """`-rcgpu_plan_only 1` (an output option like the others, rcgpu_job::options): rcgpu_encode enumerates the sequence the way image2 does (stops at the first
    gap, CLI/Input.cpp:123-317), reads the ffconcat list the reference writes for gapped sequences (CLI/Output.cpp:138-251), probes the files, settles slice
    grid and frame rate -- and stops: nothing is written and no device is looked for.  This is the part of the front end tools/fuzz/fuzz_argv.cpp drives."""
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    os.makedirs(tmp_path / "seq")
    for i in list(range(5)) + [7]:                                                          # a gap after f_000004
        (tmp_path / "seq" / ("f_%06d.dpx" % i)).write_bytes(synth.dpx_file(synth.components(64, 48, 3, 16, "film", seed=i), synth.PIX_RGB16_BE, frame_index=i))
    (tmp_path / "a.wav").write_bytes(synth.wav_file(synth.pcm_samples(4800, 2, 16), 16, 48000))
    common = ["-c:a", "flac", "-c:v", "ffv1", "-coder", "1", "-context", "1", "-f", "matroska", "-g", "1", "-level", "3", "-slicecrc", "1", "-slices", "6", "-y", "-rcgpu_plan_only", "1",
              "-f", "matroska", "out.mkv"]
    r = subprocess.run([shim, "-xerror", "-framerate", "24000/1001", "-r", "24000/1001", "-f", "image2", "-c:v", "dpx", "-start_number", "000000", "-i", "seq/f_%06d.dpx", "-i", "a.wav",
                        "-map", "0", "-map", "1"] + common, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rcgpu plan: video 5 frames 64x48 DPX/Raw/RGB/16bit/U/BE slices 3x2 fps 24000/1001 first seq/f_000000.dpx" in r.stdout
    assert "rcgpu plan: audio WAV/PCM/48kHz/16bit/2ch/S/LE 2 ch 48000 Hz 16 bit -> FLAC" in r.stdout and "rcgpu plan: output out.mkv" in r.stdout
    assert not os.path.exists(tmp_path / "out.mkv")
    (tmp_path / "list.txt").write_text("ffconcat version 1.0\n" + "".join("file 'seq/f_%06d.dpx'\r\nduration 0.041667\n" % i for i in (0, 1, 2, 3, 4, 7)))
    r = subprocess.run([shim, "-framerate", "24", "-r", "24", "-f", "concat", "-safe", "0", "-c:v", "dpx", "-i", "list.txt"] + common, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "rcgpu plan: video 6 frames 64x48" in r.stdout, r.stdout + r.stderr
    (tmp_path / "list.txt").write_text("file 'seq/f_000000.dpx'\nfile 'seq/nothing.dpx'\n")
    r = subprocess.run([shim, "-f", "concat", "-safe", "0", "-c:v", "dpx", "-i", "list.txt"] + common, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0                                                                 # (the plan stops before the files behind the first are opened)
    r = subprocess.run([shim, "-f", "concat", "-safe", "0", "-c:v", "dpx", "-i", "missing.txt"] + common, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "Error: cannot open file list" in r.stderr
    # the command line test/vulkan.sh:48-56 puts together by hand: -init_hw_device vulkan=vk:N,debug=0 in front, -hwaccel options before the input,
    # -vf hwupload -c:v ffv1_vulkan in place of -c:v ffv1 and no -context: ffv1 on HIP device N; a caller that named devices itself keeps its choice
    vk = [shim, "-xerror", "-init_hw_device", "vulkan=vk:2,debug=0", "-hwaccel", "vulkan", "-hwaccel_output_format", "vulkan", "-framerate", "24", "-f", "image2", "-c:v", "dpx",
          "-start_number", "000000", "-i", "seq/f_%06d.dpx", "-c:a", "flac", "-vf", "hwupload", "-c:v", "ffv1_vulkan", "-coder", "1", "-f", "matroska", "-g", "1", "-level", "3",
          "-slicecrc", "1", "-slices", "6", "-y", "-rcgpu_plan_only", "1", "-f", "matroska", "out.mkv"]
    r = subprocess.run(vk, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "rcgpu plan: device 2 (-init_hw_device)" in r.stdout and "rcgpu plan: video 5 frames" in r.stdout, r.stdout + r.stderr
    r = subprocess.run(vk, cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, RCGPU_DEVICES="1,1"))
    assert r.returncode == 0 and "rcgpu plan: device 1 (the caller's choice stands above -init_hw_device)" in r.stdout, r.stdout + r.stderr
    assert not os.path.exists(tmp_path / "out.mkv")
