"""`-f framemd5` (CLI/Output.cpp:312-332; rawcooked --framemd5 / --framemd5-an): the checksum FFmpeg's second output would hold for
every input frame.  What is hashed is the picture as FFmpeg's rawvideo encoder lays it out [ffmpeg-knowledge: no FFmpeg in this
image, so the layouts are unpinned]; what IS pinned here is that the device produces exactly those bytes from every file layout."""
import hashlib
import os

import numpy as np
import pytest

from rawcooked_amd import api, synth
from test_gpu_e2e import OK_LINE, SHIM, make_package, run

pytestmark = pytest.mark.gpu


def rawvideo_bytes(comp: np.ndarray, pixfmt: int) -> bytes:
    """[h, w, nc] samples (file component order R,G,B(,A) or Y) -> the bytes of FFmpeg's pix_fmt for that input."""
    bits, nc, _, be = synth.PIX_INFO[pixfmt]
    if bits == 8:
        return comp.astype(np.uint8).tobytes()                                         # rgb24 / rgba / gray
    if bits == 16:
        return comp.astype(">u2" if be else "<u2").tobytes()                           # rgb48be|le / rgba64be|le / gray16be|le
    if nc == 1:
        return comp[:, :, 0].astype("<u2").tobytes()                                   # gray10le / gray12le
    order = [1, 2, 0] + ([3] if nc == 4 else [])                                       # gbrp / gbrap: planes G, B, R(, A)
    return b"".join(comp[:, :, c].astype("<u2").tobytes() for c in order)


LAYOUTS = [(synth.PIX_RGB16_BE, 0), (synth.PIX_RGB16_LE, 0), (synth.PIX_RGB8, 0), (synth.PIX_RGBA8, 0), (synth.PIX_RGB10_FILLEDA_BE, 0),
           (synth.PIX_RGB10_FILLEDA_LE, 0), (synth.PIX_RGB12_FILLEDA_BE, 0), (synth.PIX_RGB12_PACKED_BE, 0), (synth.PIX_RGB12_PACKED_BE, synth.FLAG_VFLIP),
           (synth.PIX_RGBA10_FILLEDA_BE, 0), (synth.PIX_RGBA12_PACKED_BE, 0), (synth.PIX_RGBA16_LE, 0), (synth.PIX_Y8, 0), (synth.PIX_Y16_BE, 0),
           (synth.PIX_Y10_FILLEDA_BE, 0), (synth.PIX_Y10_FILLEDB_BE, synth.FLAG_ALTERN), (synth.PIX_Y12_PACKED_BE, 0)]


@pytest.mark.parametrize("pixfmt,flags", LAYOUTS, ids=lambda v: str(v))
def test_device_hashes_the_rawvideo_bytes_of_every_layout(built, pixfmt, flags):
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    w, h, n = 96, 40, 3                                                                 # odd line padding for 8-bit RGB: 288 B, fine; 16-bit Y: 192
    comps = [synth.components(w, h, nc, bits, "noise", seed=31 * i + pixfmt) for i in range(n)]
    packed = [synth.pack_payload(c, pixfmt, flags=flags) for c in comps]
    enc = api.Ffv1Encoder(w, h, pixfmt, packed[0][1], 2, 2, max_batch=n, flags=flags)
    try:
        enc.encode_host([p for p, _ in packed])
        sums, frame_bytes = enc.framemd5_last(n)
    finally:
        enc.close()
    # with FLAG_VFLIP pack_payload stored the lines bottom-up, and that is how FFmpeg's DPX decoder hands them on; the reference's
    # `-vf vflip` stands in front of the Matroska output and belongs to it alone: the framemd5 output hashes the lines in file order
    want = [rawvideo_bytes(c[::-1] if flags & synth.FLAG_VFLIP else c, pixfmt) for c in comps]
    assert frame_bytes == len(want[0])
    assert sums == [hashlib.md5(x).digest() for x in want]


def test_odd_width_drops_the_dpx_line_padding(built):
    """16-bit RGB, odd width: DPX pads every line to 32 bits, rawvideo does not."""
    w, h = 33, 9
    comp = synth.components(w, h, 3, 16, "film", seed=5)
    payload, line_bytes = synth.pack_payload(comp, synth.PIX_RGB16_BE)
    assert line_bytes == w * 6 + 2
    enc = api.Ffv1Encoder(w, h, synth.PIX_RGB16_BE, line_bytes, 1, 1, max_batch=1, level=1, slicecrc=0)
    try:
        enc.encode_host([payload])
        sums, frame_bytes = enc.framemd5_last(1)
    finally:
        enc.close()
    assert frame_bytes == w * h * 6 and sums[0] == hashlib.md5(comp.astype(">u2").tobytes()).digest()


def test_rawcooked_framemd5_through_the_shim(built, refbin, tmp_path):
    """`rawcooked --framemd5` puts `-f framemd5 <file>` behind the Matroska output (Output.cpp:312-332); with audio in the package the
    user adds --framemd5-an (`-an` in front of it).  For 16-bit BE RGB of even width the hashed bytes are the DPX payload itself."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 5, "film", audio=(2, 16, 48000, 12000))
    r = run([refbin, "--bin-name", SHIM, "--framemd5-an", "--check", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    lines = open(os.path.join(work, "pkg.framemd5")).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    assert head[:3] == ["#format: frame checksums", "#version: 2", "#hash: MD5"] and "#tb 0: 1/24" in head and "#dimensions 0: 64x48" in head
    assert head[-1] == "#stream#, dts,        pts, duration,     size, hash"
    rows = [l for l in lines if not l.startswith("#")]
    assert len(rows) == 5
    for i, row in enumerate(rows):
        src = open(os.path.join(work, "pkg", "img", "f_%06d.dpx" % i), "rb").read()
        assert row == "0, %10d, %10d, %8d, %8d, %s" % (i, i, 1, 64 * 48 * 6, hashlib.md5(src[2048:2048 + 64 * 48 * 6]).hexdigest())
    # without -an the audio stream is in the file too: pcm_s16le packets as FFmpeg's WAV demuxer cuts them (4096 bytes = 1024 stereo
    # 16-bit sample frames), time base 1/48000, interleaved with the video rows by time
    r = run([refbin, "--bin-name", SHIM, "--framemd5", "--check", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    lines = open(os.path.join(work, "pkg.framemd5")).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    assert "#tb 1: 1/48000" in head and "#codec_id 1: pcm_s16le" in head and "#channel_layout_name 1: stereo" in head
    rows = [l for l in lines if not l.startswith("#")]
    wav = open(os.path.join(work, "pkg", "snd.wav"), "rb").read()
    pcm = wav[api.wav_probe(wav).data_offset:][:12000 * 4]
    arows = [l for l in rows if l.startswith("1,")]
    assert len(arows) == 12 and len([l for l in rows if l.startswith("0,")]) == 5
    for j, row in enumerate(arows):
        n = min(1024, 12000 - 1024 * j)
        assert row == "1, %10d, %10d, %8d, %8d, %s" % (1024 * j, 1024 * j, n, n * 4, hashlib.md5(pcm[4096 * j:4096 * j + n * 4]).hexdigest())
    t = [(int(l.split(",")[1]) / 24.0, 0) if l.startswith("0,") else (int(l.split(",")[1]) / 48000.0, 1) for l in rows]
    assert t == sorted(t)                                                          # by time, the video frame first on a tie
