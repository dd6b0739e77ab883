"""Seeded synthetic DPX / TIFF / WAV generators.

Used by tests and bench.py: there is no network for fixtures, so every input is generated.
Content classes:
  * "film":  smooth gradient + film-grain-like noise -- the low bits are noise, so FFV1 saves only ~10-30 %
             (matches Doc/Case_study.md:236-250 of the reference for scanned film);
  * "flat":  constant colour (best case for the entropy coder);
  * "noise": uniform random samples (worst case).
All writers return bytes; nothing here depends on the oracle or on the GPU library.
"""
from __future__ import annotations

import struct
import numpy as np

# pixel layouts, numbering shared with include/rcgpu.h (RCGPU_PIX_*)
PIX_RGB8, PIX_RGB10_FILLEDA_BE, PIX_RGB10_FILLEDA_LE, PIX_RGB12_FILLEDA_BE, PIX_RGB12_FILLEDA_LE, \
    PIX_RGB16_BE, PIX_RGB16_LE, PIX_RGBA8, PIX_RGBA16_BE, PIX_RGBA16_LE, PIX_Y8, PIX_Y16_BE, PIX_Y16_LE, \
    PIX_RGB12_PACKED_BE, PIX_RGBA10_FILLEDA_BE, PIX_RGBA10_FILLEDA_LE, PIX_RGBA12_PACKED_BE, PIX_RGBA12_FILLEDA_BE, \
    PIX_RGBA12_FILLEDA_LE, PIX_Y10_FILLEDA_BE, PIX_Y10_FILLEDB_BE, PIX_Y12_PACKED_BE, PIX_EXR_RGB16 = range(23)
FLAG_VFLIP, FLAG_ALTERN = 1, 2   # RCGPU_FLAG_*

PIX_INFO = {  # pixfmt: (bits, components, bytes per pixel, big endian)
    PIX_RGB8: (8, 3, 3, False), PIX_RGB10_FILLEDA_BE: (10, 3, 4, True), PIX_RGB10_FILLEDA_LE: (10, 3, 4, False),
    PIX_RGB12_FILLEDA_BE: (12, 3, 6, True), PIX_RGB12_FILLEDA_LE: (12, 3, 6, False),
    PIX_RGB16_BE: (16, 3, 6, True), PIX_RGB16_LE: (16, 3, 6, False),
    PIX_RGBA8: (8, 4, 4, False), PIX_RGBA16_BE: (16, 4, 8, True), PIX_RGBA16_LE: (16, 4, 8, False),
    PIX_Y8: (8, 1, 1, False), PIX_Y16_BE: (16, 1, 2, True), PIX_Y16_LE: (16, 1, 2, False),
    # bit-packed DPX flavors: fields straddle bytes (bytes per pixel 0)
    PIX_RGB12_PACKED_BE: (12, 3, 0, True), PIX_RGBA10_FILLEDA_BE: (10, 4, 0, True), PIX_RGBA10_FILLEDA_LE: (10, 4, 0, False),
    PIX_RGBA12_PACKED_BE: (12, 4, 0, True), PIX_RGBA12_FILLEDA_BE: (12, 4, 8, True), PIX_RGBA12_FILLEDA_LE: (12, 4, 8, False),
    PIX_Y10_FILLEDA_BE: (10, 1, 0, True), PIX_Y10_FILLEDB_BE: (10, 1, 0, True), PIX_Y12_PACKED_BE: (12, 1, 0, True),
    PIX_EXR_RGB16: (16, 3, 0, False),     # OpenEXR scan lines: planar inside each line
}
DPX_PACKING = {  # pixfmt: DPX "packing" field (0 packed, 1 filled method A, 2 filled method B)
    PIX_RGB12_PACKED_BE: 0, PIX_RGBA12_PACKED_BE: 0, PIX_Y12_PACKED_BE: 0, PIX_Y10_FILLEDB_BE: 2,
}


def components(width: int, height: int, ncomp: int, bits: int, kind: str = "film", seed: int = 0) -> np.ndarray:
    """uint16 array [height, width, ncomp] of sample values < 2**bits."""
    rng = np.random.default_rng(seed)
    maxv = (1 << bits) - 1
    if kind == "flat":
        base = rng.integers(0, maxv + 1, size=ncomp)
        return np.broadcast_to(base.astype(np.uint16), (height, width, ncomp)).copy()
    if kind == "noise":
        return rng.integers(0, maxv + 1, size=(height, width, ncomp), dtype=np.uint16)
    if kind != "film":
        raise ValueError(kind)
    # gradient with a few soft blobs + grain whose sigma is ~1/64 of full scale (noise lives in the low bits)
    y = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    x = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    out = np.empty((height, width, ncomp), dtype=np.uint16)
    for c in range(ncomp):
        ph = rng.uniform(0, 6.28, size=3)
        sig = 0.45 + 0.25 * np.sin(3.1 * x + ph[0]) * np.cos(2.3 * y + ph[1]) + 0.15 * (x * (c + 1) / ncomp + y * 0.5) \
            + 0.05 * np.sin(40 * x + 31 * y + ph[2])
        grain = rng.normal(0.0, 1.0 / 64.0, size=(height, width)).astype(np.float32)
        v = np.clip((sig + grain) * maxv * 0.8, 0, maxv)
        out[:, :, c] = v.astype(np.uint16)
    return out


def _words_packed12(fields: np.ndarray) -> np.ndarray:
    """[n, k] 12-bit fields per row -> [n, ceil(12k/32)] words: an LSB-first bit stream cut into 32-bit words."""
    n, k = fields.shape
    nw = (k * 12 + 31) // 32
    bits = ((fields.astype(np.uint32)[:, :, None] >> np.arange(12, dtype=np.uint32)) & 1).astype(np.uint8).reshape(n, k * 12)
    bits = np.concatenate([bits, np.zeros((n, nw * 32 - k * 12), dtype=np.uint8)], axis=1).reshape(n, nw, 32)
    return (bits.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)


def _words_filled10(fields: np.ndarray, shifts) -> np.ndarray:
    """[n, k] 10-bit fields per row -> [n, ceil(k/3)] words with three fields each at `shifts`."""
    n, k = fields.shape
    nw = (k + 2) // 3
    f = np.concatenate([fields.astype(np.uint32), np.zeros((n, nw * 3 - k), dtype=np.uint32)], axis=1).reshape(n, nw, 3)
    return (f[:, :, 0] << shifts[0]) | (f[:, :, 1] << shifts[1]) | (f[:, :, 2] << shifts[2])


def pack_payload(comp: np.ndarray, pixfmt: int, dpx_line_padding: bool = True, flags: int = 0) -> tuple[bytes, int]:
    """Pack [h, w, ncomp] samples into the file layout of `pixfmt`.  Returns (payload, line_bytes)."""
    bits, ncomp, bpp, be = PIX_INFO[pixfmt]
    h, w, nc = comp.shape
    assert nc == ncomp
    if flags & FLAG_VFLIP:
        comp = comp[::-1]
    if pixfmt == PIX_EXR_RGB16:      # [y:u32][bytes:u32][B x w][G x w][R x w], little endian
        hdr = np.zeros((h, 2), dtype="<u4")
        hdr[:, 0] = np.arange(h)
        hdr[:, 1] = 6 * w
        planes = np.concatenate([comp[:, :, 2], comp[:, :, 1], comp[:, :, 0]], axis=1).astype("<u2")
        line = np.concatenate([hdr.view(np.uint8).reshape(h, 8), planes.view(np.uint8).reshape(h, 6 * w)], axis=1)
        return line.tobytes(), 8 + 6 * w
    if pixfmt in (PIX_RGB12_PACKED_BE, PIX_RGBA12_PACKED_BE, PIX_Y12_PACKED_BE):
        line = _words_packed12(comp.reshape(h, w * nc)).astype(">u4").view(np.uint8).reshape(h, -1)
    elif pixfmt in (PIX_RGBA10_FILLEDA_BE, PIX_RGBA10_FILLEDA_LE):
        line = _words_filled10(comp.reshape(h, w * nc), (22, 12, 2)).astype(">u4" if be else "<u4").view(np.uint8).reshape(h, -1)
    elif pixfmt in (PIX_Y10_FILLEDA_BE, PIX_Y10_FILLEDB_BE):
        sh = (2, 12, 22) if pixfmt == PIX_Y10_FILLEDA_BE else (0, 10, 20)
        f = comp.reshape(1, h * w) if flags & FLAG_ALTERN else comp.reshape(h, w)
        line = _words_filled10(f, sh).astype(">u4").view(np.uint8).reshape(f.shape[0], -1)
    elif pixfmt in (PIX_RGB10_FILLEDA_BE, PIX_RGB10_FILLEDA_LE):
        c = comp.astype(np.uint32)
        words = (c[:, :, 0] << 22) | (c[:, :, 1] << 12) | (c[:, :, 2] << 2)
        line = words.astype(">u4" if be else "<u4").view(np.uint8).reshape(h, w * 4)
    elif bits == 8:
        line = comp.astype(np.uint8).reshape(h, w * ncomp)
    else:
        shift = 4 if bits == 12 else 0
        line = (comp.astype(np.uint16) << shift).astype(">u2" if be else "<u2").view(np.uint8).reshape(h, w * ncomp * 2)
    line_bytes = line.shape[1]
    if dpx_line_padding and line_bytes % 4:
        pad = 4 - line_bytes % 4
        line = np.concatenate([line, np.zeros((h, pad), dtype=np.uint8)], axis=1)
        line_bytes += pad
    return line.tobytes(), (0 if flags & FLAG_ALTERN else line_bytes)


def dpx_file(comp: np.ndarray | None, pixfmt: int, fps: float = 24.0, frame_index: int = 0, big_endian_header: bool | None = None,
             trailer: bytes = b"", flags: int = 0, payload: bytes | None = None, size: tuple[int, int] | None = None) -> bytes:
    """A DPX v2.0 file (2048-byte header) whose image element uses `pixfmt`.  Either `comp` ([h, w, ncomp] samples) or a ready-made
    `payload` in that layout with its picture `size` (w, h)."""
    bits, ncomp, bpp, be = PIX_INFO[pixfmt]
    if big_endian_header is None:
        big_endian_header = be if bits > 8 else True
    e = ">" if big_endian_header else "<"
    if payload is None:
        h, w, _ = comp.shape
        # 8-bit has no endianness in the payload; >8-bit payload endianness == header endianness in DPX
        payload, _ = pack_payload(comp, pixfmt if bits == 8 else _with_endian(pixfmt, big_endian_header), True, flags)
    else:
        w, h = size
    hdr = bytearray(b"\x00" * 2048)
    hdr[0:4] = b"SDPX" if big_endian_header else b"XPDS"
    struct.pack_into(e + "I", hdr, 4, 2048)
    hdr[8:16] = b"V2.0\x00\x00\x00\x00"
    struct.pack_into(e + "I", hdr, 16, 2048 + len(payload) + len(trailer))
    struct.pack_into(e + "I", hdr, 20, 1)        # ditto key
    struct.pack_into(e + "I", hdr, 24, 1664)     # generic header size
    struct.pack_into(e + "I", hdr, 28, 384)      # industry header size
    struct.pack_into(e + "I", hdr, 32, 0)        # user data size
    name = ("frame_%06d.dpx" % frame_index).encode()
    hdr[36:36 + len(name)] = name
    hdr[136:160] = b"2026:01:01:00:00:00:UTC\x00"
    hdr[160:160 + 5] = b"rcgpu"
    if flags & FLAG_ALTERN:
        hdr[160:160 + 18] = b"Lasergraphics Inc."        # creators whose 10-bit words run across line ends (DPX.cpp:363-368)
    struct.pack_into(e + "I", hdr, 660, 0xFFFFFFFF)   # encryption key: unencrypted
    struct.pack_into(e + "H", hdr, 768, 2 if flags & FLAG_VFLIP else 0)   # orientation: 2 = bottom to top
    struct.pack_into(e + "H", hdr, 770, 1)            # number of image elements
    struct.pack_into(e + "I", hdr, 772, w)
    struct.pack_into(e + "I", hdr, 776, h)
    struct.pack_into(e + "I", hdr, 780, 0)            # data sign: unsigned
    hdr[800] = {1: 6, 3: 50, 4: 51}[ncomp]            # descriptor
    hdr[801] = 2
    hdr[802] = 2
    hdr[803] = bits
    struct.pack_into(e + "H", hdr, 804, DPX_PACKING.get(pixfmt, 1 if bits in (10, 12) else 0))   # packing
    struct.pack_into(e + "H", hdr, 806, 0)            # encoding: none
    struct.pack_into(e + "I", hdr, 808, 2048)         # offset to data
    struct.pack_into(e + "I", hdr, 812, 0)            # end-of-line padding
    struct.pack_into(e + "I", hdr, 816, 0)
    struct.pack_into(e + "f", hdr, 1724, fps)         # film: frame rate of original
    struct.pack_into(e + "I", hdr, 1712, frame_index) # frame position in sequence
    struct.pack_into(e + "f", hdr, 1940, fps)         # tv: temporal sampling rate
    return bytes(hdr) + payload + trailer


def _with_endian(pixfmt: int, be: bool) -> int:
    pairs = {PIX_RGB10_FILLEDA_BE: PIX_RGB10_FILLEDA_LE, PIX_RGB12_FILLEDA_BE: PIX_RGB12_FILLEDA_LE,
             PIX_RGB16_BE: PIX_RGB16_LE, PIX_RGBA16_BE: PIX_RGBA16_LE, PIX_Y16_BE: PIX_Y16_LE,
             PIX_RGBA10_FILLEDA_BE: PIX_RGBA10_FILLEDA_LE, PIX_RGBA12_FILLEDA_BE: PIX_RGBA12_FILLEDA_LE}
    for b, l in pairs.items():
        if pixfmt in (b, l):
            return b if be else l
    return pixfmt


def exr_file(comp: np.ndarray, fps: tuple[int, int] | None = (24, 1), trailer: bytes = b"") -> bytes:
    """An uncompressed single-part scan-line OpenEXR 2 file, channels B,G,R of type HALF (whose 16 bits are taken as they are)."""
    h, w, _ = comp.shape

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(value)) + value
    ch = b"".join(c + b"\0" + struct.pack("<IIII", 1, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack(">I", 0x762F3101) + struct.pack("<I", 2)
    hdr += attr("channels", "chlist", ch) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box)
    hdr += attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if fps:
        hdr += attr("framesPerSecond", "rational", struct.pack("<II", fps[0], fps[1]))
    hdr += b"\0"
    payload, lb = pack_payload(comp, PIX_EXR_RGB16, False)
    first = len(hdr) + 8 * h
    table = b"".join(struct.pack("<Q", first + y * lb) for y in range(h))
    return hdr + table + payload + trailer


def tiff_file(comp: np.ndarray, pixfmt: int, trailer: bytes = b"") -> bytes:
    """Baseline TIFF, single strip, tags 256,257,258,259,262,273,277,278,279,284 (+338 for RGBA)."""
    bits, ncomp, bpp, be = PIX_INFO[pixfmt]
    assert bits in (8, 16)
    e = ">" if be else "<"
    h, w, _ = comp.shape
    payload, _ = pack_payload(comp, pixfmt, False)
    tags = []
    extra = b""
    ntags = 10 + (1 if ncomp == 4 else 0)
    ifd_size = 2 + 12 * ntags + 4
    extra_off = 8 + ifd_size

    def tag(t, typ, count, value):
        tags.append(struct.pack(e + "HHI", t, typ, count) + value)

    def short(v):
        return struct.pack(e + "HH", v, 0)

    def long_(v):
        return struct.pack(e + "I", v)

    tag(256, 4, 1, long_(w))
    tag(257, 4, 1, long_(h))
    if ncomp <= 2:
        tag(258, 3, 1, short(bits))
    else:
        tag(258, 3, ncomp, long_(extra_off + len(extra)))
        extra += struct.pack(e + "H" * ncomp, *([bits] * ncomp))
    tag(259, 3, 1, short(1))
    tag(262, 3, 1, short(1 if ncomp == 1 else 2))
    data_off = extra_off + (8 if ncomp > 2 else 0)
    data_off = (data_off + 7) & ~7
    tag(273, 4, 1, long_(data_off))
    tag(277, 3, 1, short(ncomp))
    tag(278, 4, 1, long_(h))
    tag(279, 4, 1, long_(len(payload)))
    tag(284, 3, 1, short(1))
    if ncomp == 4:
        tag(338, 3, 1, short(2))
    assert len(tags) == ntags
    head = (b"MM\x00\x2a" if be else b"II\x2a\x00") + struct.pack(e + "I", 8)
    ifd = struct.pack(e + "H", ntags) + b"".join(tags) + struct.pack(e + "I", 0)
    body = head + ifd + extra
    body += b"\x00" * (data_off - len(body))
    return body + payload + trailer


def pcm_samples(n: int, channels: int, bits: int, rate: int = 48000, kind: str = "music", seed: int = 7) -> np.ndarray:
    """int32 array [n, channels] of signed samples (8-bit WAV is converted to offset binary by wav_file)."""
    rng = np.random.default_rng(seed)
    full = float((1 << (bits - 1)) - 1)
    if kind == "silence":
        return np.zeros((n, channels), dtype=np.int32)
    if kind == "noise":
        return rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(n, channels)).astype(np.int32)
    t = np.arange(n, dtype=np.float64) / rate
    out = np.empty((n, channels), dtype=np.int32)
    if kind in ("stereo", "stereo_left", "stereo_right"):
        # a centre signal in both channels plus a little of something else: what left/side, side/right and mid/side coding are for.
        # "stereo": both channels busy (mid/side wins); "stereo_left" / "stereo_right": one channel is the clean one (left/side, side/right)
        centre = 0.3 * np.sin(2 * np.pi * 330.0 * t) + 0.1 * np.sin(2 * np.pi * 1234.5 * t + 0.3) + 0.05 * np.sin(2 * np.pi * 4321.0 * t)
        wide = 0.004 * np.sin(2 * np.pi * 97.0 * t)
        for c in range(channels):
            own = (rng.random(n) - rng.random(n)) * (0.002 if kind == "stereo" or (kind == "stereo_left") == (c != 0) else 0.0)
            sig = centre + (wide if c % 2 == 0 else -wide) + own + (rng.random(n) - rng.random(n)) / full * 2.0
            out[:, c] = np.clip(np.round(sig * full), -full - 1, full).astype(np.int32)
        return out
    for c in range(channels):
        f0 = 220.0 * (1.0 + 0.37 * c)
        sig = 0.25 * np.sin(2 * np.pi * f0 * t) + 0.08 * np.sin(2 * np.pi * 3.01 * f0 * t + c)
        tpdf = (rng.random(n) - rng.random(n)) / full * 2.0
        out[:, c] = np.clip(np.round((sig + tpdf) * full), -full - 1, full).astype(np.int32)
    return out


def wav_file(samples: np.ndarray, bits: int, rate: int = 48000, extensible: bool = False, trailer_chunk: bytes = b"", float32: bool = False) -> bytes:
    n, ch = samples.shape
    if float32:
        bits = 32
        raw = (samples.astype(np.float64) / float(1 << 23)).astype("<f4").tobytes()
    elif bits == 32:
        raw = (samples.astype("<i4") * 251 + 3).astype("<i4").tobytes()      # low bits set: not a padded 24-bit signal
    elif bits == 8:
        raw = (samples + 128).astype(np.uint8).tobytes()
    elif bits == 16:
        raw = samples.astype("<i2").tobytes()
    elif bits == 24:
        b = samples.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :3]
        raw = np.ascontiguousarray(b).tobytes()
    else:
        raise ValueError(bits)
    block = ch * bits // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * block, block, bits, 22, bits, 0) + \
            struct.pack("<IHH", 3 if float32 else 1, 0, 0x0010) + bytes([0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71])
    else:
        fmt = struct.pack("<HHIIHH", 3 if float32 else 1, ch, rate, rate * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
    if len(raw) & 1:
        chunks += b"\x00"
    chunks += trailer_chunk
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks
