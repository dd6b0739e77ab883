"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Imported by tests only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "pixfmt", "num_h_slices", "num_v_slices", "ec", "context_model", "flags", "coder", "level")]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.ffv1o_config_record.restype = C.c_size_t
        L.ffv1o_encode_payload.restype = C.c_size_t
        L.ffv1o_encode_frame.restype = C.c_size_t
        L.ffv1o_trace_slice.restype = C.c_size_t
        L.ffv1o_line_bytes.restype = C.c_size_t
        L.ffv1o_payload_bytes.restype = C.c_size_t
        L.ffv1o_last_decisions.restype = C.c_uint64
        L.ffv1o_crc32.restype = C.c_uint32
        L.dpxo_padding_first_nonzero.restype = C.c_uint64
        _lib = L
    return _lib


def config_record(p: Params) -> bytes:
    buf = C.create_string_buffer(8192)
    n = lib().ffv1o_config_record(C.byref(p), buf, C.c_size_t(8192))
    return buf.raw[:n]


def encode_payload(p: Params, payload: bytes, line_bytes: int) -> bytes:
    cap = len(payload) * 2 + (1 << 16)
    out = C.create_string_buffer(cap)
    n = lib().ffv1o_encode_payload(C.byref(p), payload, C.c_size_t(line_bytes), out, C.c_size_t(cap), None)
    assert n > 0, "oracle encoder overflow"
    return C.string_at(out, n)


def decode_payload(p: Params, packet: bytes, line_bytes: int) -> bytes:
    out = C.create_string_buffer(lib().ffv1o_payload_bytes(C.byref(p), C.c_size_t(line_bytes)))
    r = lib().ffv1o_decode_payload(C.byref(p), packet, C.c_size_t(len(packet)), out, C.c_size_t(line_bytes))
    assert r == 0, f"oracle decoder error {r}"
    return out.raw


def unpack(p: Params, payload: bytes, line_bytes: int, nplanes: int) -> np.ndarray:
    planes = np.zeros((4, p.height, p.width), dtype=np.int32)
    ptrs = (C.c_void_p * 4)(*[planes[i].ctypes.data for i in range(4)])
    lib().ffv1o_unpack(C.byref(p), payload, C.c_size_t(line_bytes), ptrs)
    return planes[:nplanes]


def trace_slice(p: Params, planes: np.ndarray, sx: int, sy: int, nsamp: int):
    """-> (symbols u32[nsamp], decisions u16[], slice bytes incl. footer)"""
    full = np.zeros((4, p.height, p.width), dtype=np.int32)
    full[:planes.shape[0]] = planes
    ptrs = (C.c_void_p * 4)(*[full[i].ctypes.data for i in range(4)])
    sym = np.zeros(nsamp, dtype=np.uint32)
    dcap = nsamp * 36 + 256
    dec = np.zeros(dcap, dtype=np.uint16)
    ndec = C.c_size_t()
    rcap = nsamp * 8 + 4096
    raw = C.create_string_buffer(rcap)
    n = lib().ffv1o_trace_slice(C.byref(p), ptrs, sx, sy, sym.ctypes.data_as(C.c_void_p), dec.ctypes.data_as(C.c_void_p), C.c_size_t(dcap),
                                C.byref(ndec), raw, C.c_size_t(rcap))
    assert n > 0
    return sym, dec[:ndec.value].copy(), raw.raw[:n]


class FlacParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("channels", "sample_rate", "bits_per_sample", "block_size", "max_lpc_order")]


def flac_encode(ch, rate, bits, pcm: bytes, block_size=0, max_order=8):
    """-> (list of frames, CodecPrivate) from the scalar oracle."""
    L = lib()
    L.flaco_encode.restype = C.c_long
    L.flaco_codec_private.restype = C.c_size_t
    L.flaco_default_block_size.restype = C.c_uint32
    p = FlacParams(ch, rate, bits, block_size, max_order)
    n = len(pcm) // (ch * bits // 8)
    out = C.create_string_buffer(len(pcm) * 2 + 65536)
    fs = (C.c_uint32 * (n // 16 + 16))()
    nf = L.flaco_encode(C.byref(p), pcm, C.c_uint64(len(pcm)), out, C.c_size_t(len(out)), fs, C.c_size_t(len(fs)))
    assert nf >= 0, nf
    frames, off = [], 0
    raw = out.raw
    for i in range(nf):
        frames.append(raw[off:off + fs[i]])
        off += fs[i]
    md5 = C.create_string_buffer(16)
    spcm = pcm[:n * ch * (bits // 8)]
    if bits == 8:
        spcm = (np.frombuffer(spcm, dtype=np.uint8).astype(np.int16) - 128).astype(np.int8).tobytes()
    L.flaco_md5(spcm, C.c_size_t(len(spcm)), md5)
    cp = C.create_string_buffer(64)
    sizes = [fs[i] for i in range(nf)] or [0]
    k = L.flaco_codec_private(C.byref(p), C.c_uint64(n), min(sizes), max(sizes), md5, cp)
    return frames, cp.raw[:k]


def flac_decode(ch, rate, bits, data: bytes, pcm_len: int) -> bytes:
    L = lib()
    L.flaco_decode.restype = C.c_longlong
    p = FlacParams(ch, rate, bits, 0, 8)
    out = C.create_string_buffer(pcm_len + 64)
    r = L.flaco_decode(C.byref(p), data, C.c_size_t(len(data)), out, C.c_size_t(len(out)))
    assert r >= 0, f"oracle FLAC decoder error {r}"
    return out.raw[:r]


def dpx_padding_first_nonzero(payload: bytes, width: int, height: int, bit_depth: int, components: int, big_endian: bool, packing: int, altern: bool) -> int:
    """DPX.cpp:501-608 restated (oracle/dpx_oracle.c); packing 0 Packed, 1 FilledA, 2 FilledB.  -> offset or 2**64 - 1."""
    return lib().dpxo_padding_first_nonzero(payload, C.c_uint64(len(payload)), width, height, bit_depth, components, int(big_endian), packing, int(altern))
