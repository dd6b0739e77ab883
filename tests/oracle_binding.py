"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Imported by tests only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class StreamExt(C.Structure):
    """ffv1o_stream_ext: a stream laid out as freely as parameters::Parse reads one (oracle/ffv1_oracle.h)."""
    _fields_ = [("version", C.c_uint32), ("micro_version", C.c_uint32), ("custom_transitions", C.c_uint32), ("one_state", C.c_uint8 * 256),
                ("set_count", C.c_uint32), ("levels", C.c_uint8 * 128 * 5 * 8), ("states_coded", C.c_uint32 * 8),
                ("initial_states", C.c_void_p * 8), ("set_index", C.c_uint32 * 3), ("intra", C.c_uint32), ("alt_slices", C.c_uint32),
                ("set_index_alt", C.c_uint32 * 3)]


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "pixfmt", "num_h_slices", "num_v_slices", "ec", "context_model", "flags", "coder", "level")] + [("ext", C.POINTER(StreamExt))]


def levels_from_runs(runs):
    """run lengths of equal levels (what the record carries, FFV1_Parameters.cpp:222-253) -> the level of every |difference| 0..127"""
    out = []
    for level, r in enumerate(runs):
        out += [level] * r
    assert len(out) == 128, (runs, len(out))
    return out


def stream_ext(version=3, sets=None, set_index=(0, 0, 0), one_state=None, initial_states=None, micro_version=4, intra=1, set_index_alt=None):
    """sets: list of table sets, each five lists of run lengths; initial_states: {set: bytes(context_count * 32)}.  The returned object keeps
    the buffers it points to alive (`_keep`)."""
    e = StreamExt()
    e.version, e.micro_version, e.intra = version, micro_version, intra
    if one_state is not None:
        e.custom_transitions = 1
        for i, v in enumerate(one_state):
            e.one_state[i] = v
    e.set_count = len(sets)
    for i, st in enumerate(sets):
        for j, runs in enumerate(st):
            for k, lv in enumerate(levels_from_runs(runs)):
                e.levels[i][j][k] = lv
    for g in range(3):
        e.set_index[g] = set_index[g]
    if set_index_alt is not None:                     # the slices with odd sx + sy say so instead: the index is a per-slice field
        e.alt_slices = 1
        for g in range(3):
            e.set_index_alt[g] = set_index_alt[g]
    e._keep = []
    for i, b in (initial_states or {}).items():
        assert len(b) == lib().ffv1o_ext_context_count(C.byref(e), i) * 32
        buf = C.create_string_buffer(bytes(b), len(b))
        e._keep.append(buf)
        e.states_coded[i] = 1
        e.initial_states[i] = C.cast(buf, C.c_void_p)
    return e


def stream_ext_from_vector(v, golden_dir):
    """the `ext` entry of a tests/golden/vectors.json ffv1_ext vector -> StreamExt"""
    import os
    kw = dict(v["ext"])
    if "initial_states" in kw:
        kw["initial_states"] = {int(i): open(os.path.join(golden_dir, fn), "rb").read() for i, fn in kw["initial_states"].items()}
    return stream_ext(**kw)


def with_ext(p: Params, e: StreamExt) -> Params:
    q = Params(p.width, p.height, p.pixfmt, p.num_h_slices, p.num_v_slices, p.ec, p.context_model, p.flags, p.coder, p.level)
    q.ext = C.pointer(e)
    q._keep = e
    return q


def decode_stream(p: Params, record: bytes, packet: bytes, line_bytes: int):
    """ffv1o_decode_stream: everything about the stream from the record / the packet itself.  -> (error code, payload)"""
    out = C.create_string_buffer(lib().ffv1o_payload_bytes(C.byref(p), C.c_size_t(line_bytes)))
    r = lib().ffv1o_decode_stream(C.byref(p), record, C.c_size_t(len(record)), packet, C.c_size_t(len(packet)), out, C.c_size_t(line_bytes))
    return r, out.raw


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
    buf = C.create_string_buffer(1 << 22)                 # coded initial states: up to 32768 x 32 symbols a set
    n = lib().ffv1o_config_record(C.byref(p), buf, C.c_size_t(len(buf)))
    return buf.raw[:n]


def encode_payload(p: Params, payload: bytes, line_bytes: int) -> bytes:
    cap = len(payload) * (2 if not p.ext else 6) + (1 << 16)      # (a stream with adverse initial states or transitions can expand a flat picture several times)
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
