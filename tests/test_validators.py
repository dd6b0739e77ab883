"""CPU: the two specification-level validators (tests/rfc9043_validator.py, tests/mkv_validator.py) on everything the CPU suite can
produce -- golden streams blessed by the real reference, oracle streams over the option surface, files written by the muxer through
both of its block paths -- and their own negative controls.  Stock FFmpeg / mkvalidator do not exist in this environment; these stand
where they would."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import pytest

import flac_validator
import mkv_validator
import oracle_binding as ob
import rfc9043_validator as rfc
from rawcooked_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))


@pytest.mark.parametrize("v", [v for v in VEC["ffv1"] if v["config_record"]], ids=lambda v: v["name"])
def test_golden_streams_are_structurally_valid_ffv1(built, v):
    packets = [open(os.path.join(G, fr["packet"]), "rb").read() for fr in v["frames"]]
    r = rfc.validate_stream(bytes.fromhex(v["config_record"]), packets, v["width"], v["height"])
    assert (r.num_h_slices, r.num_v_slices) == (v["num_h"], v["num_v"]) and r.coder_type == v["coder"]


def _repacked(v, pixels):
    """The decoder's pixels in the file layout of the vector (rawcooked_amd/synth.py: numpy, no oracle, no library)."""
    import numpy as np
    comp = np.array(pixels, dtype=np.int64)
    bits = synth.PIX_INFO[v["pixfmt"]][0]
    assert comp.min() >= 0 and comp.max() < (1 << bits), "a decoded component leaves the sample range"
    pl, _ = synth.pack_payload(comp.astype(np.uint16), v["pixfmt"], dpx_line_padding=v["name"].startswith("dpx"), flags=v["flags"])
    return pl


@pytest.mark.parametrize("v", [v for v in VEC["ffv1"] if v["config_record"]], ids=lambda v: v["name"])
def test_golden_packets_decode_to_their_sources_by_the_rfc_alone(built, v):
    """Sample-level conformance without any of this repository's codec code: every golden packet is decoded by tests/rfc9043_validator.py
    -- border, predictor, contexts, RCT, symbols written from RFC 9043's text -- and must give the payload the reference (`rawcooked -y`)
    and the oracle rebuilt from it.  Three decoders, one answer; what remains unpinned is FFmpeg itself."""
    r = rfc.parse_record(bytes.fromhex(v["config_record"]))
    for fr in v["frames"]:
        packet = open(os.path.join(G, fr["packet"]), "rb").read()
        pixels = rfc.decode_frame(r, packet, v["width"], v["height"])
        assert _repacked(v, pixels) == open(os.path.join(G, fr["payload"]), "rb").read(), fr["packet"]


def test_the_rfc_decoder_is_not_fooled(built):
    """Negative control: one flipped bit inside a slice's coded samples changes what the RFC decoder rebuilds (or trips its checks)."""
    v = next(x for x in VEC["ffv1"] if x["name"] == "dpx_rgb16be_64x48")
    r = rfc.parse_record(bytes.fromhex(v["config_record"]))
    r.ec = 0                                                                          # look past the CRC: the samples themselves must differ
    packet = bytearray(open(os.path.join(G, v["frames"][0]["packet"]), "rb").read())
    want = open(os.path.join(G, v["frames"][0]["payload"]), "rb").read()
    # with ec cleared the footer is 3 bytes: rebuild the packet without CRCs to keep the walk from the tail valid
    rr = rfc.parse_record(bytes.fromhex(v["config_record"]))
    slices = rfc.split_slices(rr, bytes(packet))
    a, b = slices[0]
    packet[a + (b - a) // 2] ^= 0x04
    fixed = bytearray()
    for (x, y) in slices:
        fixed += packet[x:y - 5]                                                     # drop error_status + CRC, keep slice_size
    try:
        got = _repacked(v, rfc.decode_frame(r, bytes(fixed), v["width"], v["height"]))
    except AssertionError:
        return
    assert got != want


FLAC_WORDS = json.load(open(os.path.join(G, "flac_coefficients.json")))


def words_diff(name, got, want):
    """Where two lists of predictor words (flac_validator.predictor_words) part: names the frame, channel and field."""
    assert len(got) == len(want), f"{name}: {len(got)} frames, the file holds {len(want)}"
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[0] == w[0], f"{name}: frame {k}: channel assignment {g[0]}, pinned {w[0]}"
        for c, (gs, ws) in enumerate(zip(g[1], w[1])):
            for field, a, b in zip(("type", "order", "precision", "shift", "coefficients", "partition order"), gs, ws):
                assert a == b, f"{name}: frame {k} channel {c}: {field} {a}, pinned {b}"


@pytest.mark.parametrize("v", VEC["flac"], ids=lambda v: v["name"])
def test_golden_flac_streams_decode_by_the_format_specification_alone(built, v):
    """tests/flac_validator.py -- frame header, CRC-8 / CRC-16, subframes, Rice residuals, stereo decorrelation written from the FLAC format
    description -- rebuilds the PCM of every golden stream (which the reference's vendored libFLAC and the oracle's decoder rebuild
    too), the STREAMINFO MD5 is that of the signed samples, and the predictor words are the pinned ones."""
    import hashlib
    frames = open(os.path.join(G, v["frames"]), "rb").read()
    pcm = open(os.path.join(G, v["pcm"]), "rb").read()
    got, infos = flac_validator.parse_stream(frames, v["channels"], v["bits"], v["rate"])
    assert got == pcm
    signed = bytes(b ^ 0x80 for b in pcm) if v["bits"] == 8 else pcm                 # FLAC hashes the signed samples; 8-bit WAV is unsigned
    assert hashlib.md5(signed).digest() == bytes.fromhex(v["codec_private"])[-16:]
    words_diff(v["name"], flac_validator.predictor_words(infos), FLAC_WORDS[v["name"]])


@pytest.mark.parametrize("v", VEC["flac"], ids=lambda v: v["name"])
def test_oracle_flac_predictor_words_are_the_pinned_ones(built, v):
    """The scalar oracle, compiled here and now, still chooses the coefficient words of tests/golden/flac_coefficients.json (a compiler
    that moved a rounding in Levinson-Durbin or the quantiser shows up as a named coefficient, not as a frame of another size)."""
    pcm = open(os.path.join(G, v["pcm"]), "rb").read()
    frames, _ = ob.flac_encode(v["channels"], v["rate"], v["bits"], pcm, 0, 8)
    _, infos = flac_validator.parse_stream(b"".join(frames), v["channels"], v["bits"], v["rate"])
    words_diff(v["name"], flac_validator.predictor_words(infos), FLAC_WORDS[v["name"]])


def test_flac_validator_negative_controls(built):
    v = VEC["flac"][0]
    frames = bytearray(open(os.path.join(G, v["frames"]), "rb").read())
    frames[40] ^= 0x01
    with pytest.raises(AssertionError):
        flac_validator.parse_stream(bytes(frames), v["channels"], v["bits"], v["rate"])
    words = json.loads(json.dumps(FLAC_WORDS[v["name"]]))
    lpc = next(s for f in words for s in f[1] if s[0] == "lpc")
    lpc[4][0] += 1
    with pytest.raises(AssertionError, match="coefficients"):
        words_diff(v["name"], words, FLAC_WORDS[v["name"]])


@pytest.mark.parametrize("pixfmt,w,h,nh,nv,slicecrc,context,coder", [
    (synth.PIX_RGB16_BE, 200, 120, 3, 2, 0, 1, 1), (synth.PIX_RGB16_BE, 200, 120, 3, 2, 1, 0, 1), (synth.PIX_RGB10_FILLEDA_BE, 130, 67, 3, 3, 0, 0, 1),
    (synth.PIX_RGBA16_LE, 64, 48, 2, 2, 1, 1, 2), (synth.PIX_Y16_BE, 80, 40, 2, 2, 1, 1, 1), (synth.PIX_RGB8, 257, 131, 4, 4, 1, 2, 1)])
def test_oracle_streams_over_the_option_surface_are_valid(built, pixfmt, w, h, nh, nv, slicecrc, context, coder):
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    p = ob.Params(w, h, pixfmt, nh, nv, slicecrc, context, 0, coder)
    packets = []
    for i in range(2):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, "film" if i else "noise", seed=9 + i), pixfmt, True)
        packets.append(ob.encode_payload(p, pl, line_bytes))
    r = rfc.validate_stream(ob.config_record(p), packets, w, h)
    assert r.ec == slicecrc and r.bits_per_raw_sample == bits and r.alpha_plane == (nc == 4) and r.colorspace_type == (0 if nc == 1 else 1)
    assert len(r.context_count) == 2 and r.context_count[1] == (338 if context == 2 else 6561 if bits <= 8 else 5063)      # 4.9.3


def test_ffv1_validator_negative_controls(built):
    w, h, pixfmt = 96, 64, synth.PIX_RGB16_BE
    p = ob.Params(w, h, pixfmt, 2, 2, 1, 1)
    pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=1), pixfmt, True)
    rec, pk = ob.config_record(p), ob.encode_payload(p, pl, line_bytes)
    rfc.validate_stream(rec, [pk], w, h)
    with pytest.raises(AssertionError, match="crc_parity"):
        rfc.parse_record(rec[:-1] + bytes([rec[-1] ^ 1]))
    with pytest.raises(AssertionError):
        rfc.validate_stream(rec, [pk[:-1] + bytes([pk[-1] ^ 0x10])], w, h)          # slice CRC
    with pytest.raises(AssertionError):
        rfc.validate_stream(rec, [pk + pk], w, h)                                    # eight slices in a 2x2 grid
    bad = bytearray(pk); bad[len(pk) - 7] ^= 0x01                                    # slice_size of the last slice
    with pytest.raises(AssertionError):
        rfc.validate_stream(rec, [bytes(bad)], w, h)


@pytest.mark.parametrize("mode", ["write", "mapped", "tmpfs", "pwrite"])
def test_muxer_files_are_valid_matroska_and_the_reference_reads_them(built, tmp_path, monkeypatch, mode):
    """The same package through the muxer's block paths -- buffered writes; payloads copied into a mapping of the file (forced on
    whatever file system the test runs on, and on tmpfs, where the muxer chooses it itself, allocates the pages ahead and the writers
    pre-fault their range); payloads through pwrite(): identical blocks in the file, valid by the Matroska rules, and -- where the real
    reference is built -- rebuilt bit-exactly by it."""
    import shutil
    import tempfile
    if mode == "pwrite":
        monkeypatch.setenv("RCGPU_MKV_NO_MMAP", "1")
    if mode == "mapped":
        monkeypatch.setenv("RCGPU_MKV_MMAP", "1")
    work = str(tmp_path)
    if mode == "tmpfs":
        if not os.path.isdir("/dev/shm"):
            pytest.skip("no /dev/shm")
        work = tempfile.mkdtemp(prefix="rcgpu_test_", dir="/dev/shm")
    try:
        _muxer_paths(work, mode)
    finally:
        if mode == "tmpfs":
            shutil.rmtree(work, ignore_errors=True)


def _muxer_paths(work, mode):
    os.makedirs(work + "/pkg/img")
    w, h, pixfmt, n = 72, 40, synth.PIX_RGB16_BE, 7
    files = []
    for i in range(n):
        fn = work + "/pkg/img/f_%06d.dpx" % i
        open(fn, "wb").write(synth.dpx_file(synth.components(w, h, 3, 16, "film", seed=i), pixfmt, frame_index=i))
        files.append(fn)
    info = api.dpx_probe(open(files[0], "rb").read())
    nh, nv = api.slices_to_grid(info.slices)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    packets = [ob.encode_payload(p, open(fn, "rb").read()[info.data_offset:info.data_offset + info.data_size], info.line_bytes) for fn in files]
    ref = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "rawcooked")
    have_ref = os.path.exists(ref)
    path = work + "/pkg.mkv"
    if have_ref:
        r = subprocess.run([ref, "--hash", "--no-check-padding", "-d", "-y", "pkg"], cwd=work, capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
        assert r.returncode == 0, r.stderr
    L = api.lib()
    mux = api.MkvMuxer(path)
    tv = mux.add_video(ob.config_record(p), w, h, 24, 1)
    if have_ref:
        mux.add_attachment("RAWcooked reversibility data", open(work + "/pkg.rawcooked_reversibility_data", "rb").read())
    mux.add_tag(tv, "ENCODER", "rcgpu test")
    mux.begin()
    if mode != "write":
        assert L.rcgpu_mkv_expect(mux.h, sum(len(x) for x in packets) + 4096, n) == 0
    jobs = []
    for i, pk in enumerate(packets):
        if mode == "write":
            mux.write_block(tv, i * 10 ** 9 // 24, pk)
        else:
            dst, off = C.c_void_p(), C.c_uint64()
            assert L.rcgpu_mkv_reserve_block(mux.h, tv, i * 10 ** 9 // 24, len(pk), 1, C.byref(dst), C.byref(off)) == 0
            assert bool(dst.value) == (mode in ("mapped", "tmpfs"))
            jobs.append((dst.value, off.value, pk))

    def fill(part):
        for dst, off, pk in part:
            if dst:
                assert L.rcgpu_mkv_copy_in(mux.h, dst, pk, len(pk)) == 0
            else:
                assert L.rcgpu_mkv_fill(mux.h, off, pk, len(pk)) == 0
    ths = [threading.Thread(target=fill, args=(jobs[k::3],)) for k in range(3)]
    [t.start() for t in ths]; [t.join() for t in ths]
    mux.close()
    rep = mkv_validator.validate(path)
    assert rep["tracks"][1] == {"type": 1, "codec": "V_FFV1", "blocks": n} and rep["cues"] == n and rep["trailing_bytes"] == 0
    data = open(path, "rb").read()
    at = 0
    for pk in packets:                                     # every packet is in the file, whole and in order
        at = data.index(pk, at) + len(pk)
    if have_ref:
        r = subprocess.run([ref, "--check", "pkg.mkv"], cwd=work, capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
        assert r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout, r.stdout + r.stderr


def test_mkv_validator_negative_controls(built, tmp_path):
    path = str(tmp_path / "a.mkv")
    mux = api.MkvMuxer(path)
    tv = mux.add_video(b"\x01\x02", 64, 48, 24, 1)
    ta = mux.add_audio(b"fLaC" + bytes(38), 2, 48000, 16)
    mux.begin()
    for i in range(4):
        mux.write_block(ta, i * 96000000, b"A" * 50)
        mux.write_block(tv, i * 10 ** 9 // 24, os.urandom(2 << 20) if i == 2 else b"V" * 90)
    mux.close()
    rep = mkv_validator.validate(path)
    assert rep["tracks"][1]["blocks"] == 4 and rep["tracks"][2]["blocks"] == 4
    good = open(path, "rb").read()
    seg = good.index(bytes.fromhex("18538067"))
    bad = bytearray(good); bad[seg + 11] ^= 0x01                               # Segment size no longer matches
    (tmp_path / "b.mkv").write_bytes(bad)
    with pytest.raises(AssertionError):
        mkv_validator.validate(str(tmp_path / "b.mkv"))
    blk = good.index(b"V" * 90) - 4                                           # track number byte of a SimpleBlock
    bad = bytearray(good); bad[blk] = 0x80 | 9
    (tmp_path / "c.mkv").write_bytes(bad)
    with pytest.raises(AssertionError, match="unknown track"):
        mkv_validator.validate(str(tmp_path / "c.mkv"))
    (tmp_path / "d.mkv").write_bytes(good + b"junk after the segment")
    with pytest.raises(AssertionError):
        mkv_validator.validate(str(tmp_path / "d.mkv"))


_FULL_FS_SCRIPT = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[2])
from rawcooked_amd import api
L = api.lib()
mux = api.MkvMuxer(os.path.join(sys.argv[1], "out.mkv"))
tv = mux.add_video(b"\x00" * 8, 64, 48, 24, 1)
mux.begin()
block = bytes(range(256)) * 4096                        # 1 MiB
assert L.rcgpu_mkv_expect(mux.h, 64 << 20, 64) == 0      # the file system holds 8 MiB
errors = 0
for i in range(24):
    dst, off = C.c_void_p(), C.c_uint64()
    if L.rcgpu_mkv_reserve_block(mux.h, tv, i * 10 ** 9 // 24, len(block), 1, C.byref(dst), C.byref(off)) != 0:
        errors += 1; break
    r = L.rcgpu_mkv_copy_in(mux.h, dst, block, len(block)) if dst.value else L.rcgpu_mkv_fill(mux.h, off, block, len(block))
    if r != 0:
        errors += 1
        assert "No space left" in api.last_error() or "failed" in api.last_error(), api.last_error()
        break
assert errors == 1, "24 MiB went into an 8 MiB file system without an error"
print("full file system reported:", api.last_error())
"""


def test_a_full_tmpfs_is_an_error_code_not_a_sigbus(built, tmp_path):
    """The muxer maps its output on tmpfs and allocates pages ahead with fallocate(); when the file system is full, the payloads must
    come back as a write error (the job then exits with a code and removes the partial file), not as a SIGBUS inside a memcpy."""
    fs = str(tmp_path / "fs")
    os.makedirs(fs)
    if subprocess.run(["mount", "-t", "tmpfs", "-o", "size=8m", "none", fs], capture_output=True).returncode != 0:
        pytest.skip("cannot mount a small tmpfs here")
    try:
        r = subprocess.run([sys.executable, "-c", _FULL_FS_SCRIPT, fs, os.path.dirname(HERE)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, "exit %d (-7 = SIGBUS)\n%s\n%s" % (r.returncode, r.stdout, r.stderr)
        assert "full file system reported" in r.stdout
    finally:
        subprocess.run(["umount", fs], capture_output=True)
