"""ctypes binding of librcgpu.so (include/rcgpu.h).

Python is only the harness language here (tests, bench.py, __graft_entry__): the product is the C-ABI library.
The names mirror the reference's interface for this path -- `output.Process()` (Source/CLI/Output.cpp:22) hands a
list of `stream`s and `attachment`s (Output.h:21-39) to the encoder; `Output` below does the same through
`rcgpu_encode`.  Nothing in this module imports the oracle, and there is no CPU fallback: when the HIP library is
missing or no device is visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librcgpu.so")
if os.environ.get("RCGPU_LIB"):      # the measuring tools load the timing build (make -C rawcooked_amd/csrc timing) this way; nothing else does
    LIB_PATH = os.path.abspath(os.environ["RCGPU_LIB"])


class RcgpuError(RuntimeError):
    code = 0


class RcgpuUnsupported(RcgpuError):
    """RCGPU_FFV1_UNSUPPORTED: a valid stream the device decoder does not take -- the caller's own decoder does"""
    code = 20


class ImageInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixfmt", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("data_offset", C.c_uint64), ("data_size", C.c_uint64), ("line_bytes", C.c_uint32), ("slices", C.c_uint32),
                ("framerate", C.c_double), ("flavor", C.c_char * 64), ("flags", C.c_uint32)]


class AudioInfo(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("sample_rate", C.c_uint32), ("bits_per_sample", C.c_uint32), ("block_align", C.c_uint32),
                ("data_offset", C.c_uint64), ("data_size", C.c_uint64), ("flavor", C.c_char * 64), ("format_tag", C.c_uint32)]


class Ffv1Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixfmt", C.c_uint32), ("line_bytes", C.c_uint32),
                ("num_h_slices", C.c_uint32), ("num_v_slices", C.c_uint32), ("slicecrc", C.c_uint32), ("context", C.c_uint32),
                ("max_batch", C.c_uint32), ("device", C.c_int), ("segments", C.c_uint32), ("flags", C.c_uint32), ("coder", C.c_uint32), ("level", C.c_uint32),
                ("rc_span", C.c_uint32), ("slice_buffer_div", C.c_uint32)]


class Ffv1StreamInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("version", "micro_version", "coder_type", "colorspace_type", "bits_per_raw_sample", "chroma_planes", "alpha_plane",
                                          "num_h_slices", "num_v_slices", "quant_table_set_count", "ec", "intra", "quant_table_set_index_count")] + \
               [("quant_table_set_index", C.c_uint32 * 3), ("context_count", C.c_uint32 * 8), ("states_coded", C.c_uint32 * 8)]


READ_FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t)
PLACE_PACKET_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t)
PACKET_DONE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t)
LOCATE_FRAME_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t)


class SequenceIo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("read_frame", READ_FRAME_FN), ("place_packet", PLACE_PACKET_FN), ("packet_done", PACKET_DONE_FN), ("user", C.c_void_p),
                ("locate_frame", LOCATE_FRAME_FN)]


class SequenceOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device_first", C.c_int), ("device_count", C.c_int), ("batch", C.c_uint32), ("readers", C.c_uint32), ("writers", C.c_uint32),
                ("in_ring_frames", C.c_uint32), ("out_ring_bytes", C.c_uint64), ("lanes_per_device", C.c_uint32),
                ("copy_streams", C.c_uint32), ("device_aliases", C.c_uint32), ("frames_pinned", C.c_uint32), ("run_on", C.c_uint32), ("numa", C.c_uint32)]


class SequenceStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("first_packet_seconds", C.c_double), ("prepare_seconds", C.c_double), ("device_busy_seconds", C.c_double),
                ("frames", C.c_uint64), ("payload_bytes", C.c_uint64), ("packet_bytes", C.c_uint64), ("batches", C.c_uint64),
                ("batch_frames", C.c_uint32), ("devices", C.c_uint32), ("readers", C.c_uint32), ("writers", C.c_uint32),
                ("steady_frames_per_second", C.c_double), ("reads_done_seconds", C.c_double), ("last_batch_seconds", C.c_double),
                ("upload_wait_seconds", C.c_double), ("h2d_span_seconds", C.c_double), ("read_call_seconds", C.c_double), ("write_call_seconds", C.c_double),
                ("host_groups", C.c_uint32), ("lane_device", C.c_int32 * 16), ("lane_numa_node", C.c_int32 * 16), ("lane_pinned_node", C.c_int32 * 16)]


class FlacConfig(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("sample_rate", C.c_uint32), ("bits_per_sample", C.c_uint32), ("block_size", C.c_uint32),
                ("max_lpc_order", C.c_uint32), ("device", C.c_int)]


class _Stream(C.Structure):
    _fields_ = [("path_or_template", C.c_char_p), ("start_number", C.c_char_p), ("filelist", C.c_char_p), ("flavor", C.c_char_p),
                ("framerate", C.c_char_p), ("slices", C.c_uint32), ("vflip", C.c_int)]


class _Attachment(C.Structure):
    _fields_ = [("path_in", C.c_char_p), ("name_out", C.c_char_p)]


class _Job(C.Structure):
    _fields_ = [("streams", C.POINTER(_Stream)), ("n_streams", C.c_size_t), ("attachments", C.POINTER(_Attachment)), ("n_attachments", C.c_size_t),
                ("reversibility_path", C.c_char_p), ("output_path", C.c_char_p), ("framemd5_path", C.c_char_p),
                ("options", C.POINTER(C.c_char_p)), ("n_options", C.c_size_t), ("device_first", C.c_int), ("device_count", C.c_int)]


# every symbol include/rcgpu.h declares: name -> (restype, argtypes)
_VP, _SZ, _U8P = C.c_void_p, C.c_size_t, C.c_char_p
SYMBOLS = {
    "rcgpu_version": (C.c_char_p, []),
    "rcgpu_last_error": (C.c_char_p, []),
    "rcgpu_device_count": (C.c_int, []),
    "rcgpu_encode": (C.c_int, [C.POINTER(_Job)]),
    "rcgpu_main_ffmpeg_argv": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "rcgpu_dpx_probe": (C.c_int, [_U8P, _SZ, C.POINTER(ImageInfo)]),
    "rcgpu_tiff_probe": (C.c_int, [_U8P, _SZ, C.POINTER(ImageInfo)]),
    "rcgpu_exr_probe": (C.c_int, [_U8P, _SZ, C.POINTER(ImageInfo)]),
    "rcgpu_wav_probe": (C.c_int, [_U8P, _SZ, C.POINTER(AudioInfo)]),
    "rcgpu_pixfmt_from_flavor": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32)]),
    "rcgpu_reference_slices": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rcgpu_slices_to_grid": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rcgpu_ffv1_create": (C.c_int, [C.POINTER(Ffv1Config), C.POINTER(_VP)]),
    "rcgpu_ffv1_destroy": (None, [_VP]),
    "rcgpu_ffv1_config_record": (_SZ, [_VP, _VP, _SZ]),
    "rcgpu_ffv1_max_packet_bytes": (_SZ, [_VP]),
    "rcgpu_ffv1_device_bytes_per_frame": (C.c_uint64, [_VP, C.c_int]),
    "rcgpu_ffv1_batch_intervals": (C.c_int, [_VP, C.POINTER(C.c_float), C.c_int]),
    "rcgpu_ffv1_encode_device": (C.c_int, [_VP, C.POINTER(_VP), C.c_uint32, _VP, _SZ, _VP, _VP]),
    "rcgpu_ffv1_encode_host": (C.c_int, [_VP, C.POINTER(_VP), C.c_uint32, C.POINTER(_VP), C.POINTER(_SZ)]),
    "rcgpu_ffv1_last_error_flags": (C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    "rcgpu_ffv1_set_run_on": (C.c_int, [_VP, C.c_int]),
    "rcgpu_ffv1_join": (C.c_int, [_VP, _VP]),
    "rcgpu_ffv1_run_on": (C.c_int, [_VP]),
    "rcgpu_sequence_plan": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rcgpu_sequence_plan_lanes": (C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "rcgpu_release_device_streams": (None, []),
    "rcgpu_ffv1_encode_sequence": (C.c_int, [C.POINTER(Ffv1Config), C.c_uint64, C.POINTER(SequenceIo), C.POINTER(SequenceOptions), C.POINTER(SequenceStats), _VP, C.POINTER(_SZ)]),
    "rcgpu_ffv1_encode_sequence_memory": (C.c_int, [C.POINTER(Ffv1Config), C.POINTER(_VP), C.c_uint64, C.c_uint64, C.POINTER(_VP), C.c_uint64, _SZ, C.POINTER(C.c_uint64),
                                          C.POINTER(SequenceOptions), C.POINTER(SequenceStats), _VP, C.POINTER(_SZ)]),
    "rcgpu_mkv_expect": (C.c_int, [_VP, C.c_uint64, C.c_uint64]),
    "rcgpu_mkv_reserve_block": (C.c_int, [_VP, C.c_int, C.c_uint64, _SZ, C.c_int, C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    "rcgpu_mkv_copy_in": (C.c_int, [_VP, _VP, _VP, _SZ]),
    "rcgpu_mkv_fill": (C.c_int, [_VP, C.c_uint64, _VP, _SZ]),
    "rcgpu_ffv1_framemd5_last": (C.c_int, [_VP, C.c_uint32, _VP, C.POINTER(C.c_uint64)]),
    "rcgpu_ffv1_last_kernel_times": (C.c_int, [_VP, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    "rcgpu_ffv1_last_kernel_launches": (C.c_int, [_VP, C.c_int]),
    "rcgpu_ffv1_last_stats": (C.c_int, [_VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rcgpu_ffv1_decoder_create": (C.c_int, [C.POINTER(Ffv1Config), C.POINTER(_VP)]),
    "rcgpu_ffv1_decoder_destroy": (None, [_VP]),
    "rcgpu_ffv1_decoder_decode_device": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(_VP), C.POINTER(C.c_uint32), _VP]),
    "rcgpu_md5_host_batch": (C.c_int, [C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32, _U8P, C.c_int]),
    "rcgpu_analysis_host_batch": (C.c_int, [C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32, _VP, _VP, _VP, C.c_int]),
    "rcgpu_ffv1_decoder_decode_host": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(_VP)]),
    "rcgpu_ffv1_decoder_decode_keep": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32]),
    "rcgpu_ffv1_decoder_decode_keep_hint": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32]),
    "rcgpu_ffv1_decoder_decode_keep_hint_file": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32]),
    "rcgpu_ffv1_decoder_decode_keep_adopt": (C.c_int, [_VP]),
    "rcgpu_ffv1_decoder_decode_keep_fd": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32]),
    "rcgpu_ffv1_decoder_kept_to_host": (C.c_int, [_VP, C.c_uint32, _VP]),
    "rcgpu_ffv1_decoder_verify_kept": (C.c_int, [_VP, _VP, C.c_uint32, _VP]),
    "rcgpu_ffv1_decoder_verify_kept_begin": (C.c_int, [_VP, _VP, C.c_uint32]),
    "rcgpu_ffv1_decoder_verify_kept_end": (C.c_int, [_VP, _VP]),
    "rcgpu_ffv1_config_from_record": (C.c_int, [_U8P, _SZ, C.POINTER(Ffv1Config)]),
    "rcgpu_ffv1_config_from_stream": (C.c_int, [_U8P, _SZ, _U8P, _SZ, C.POINTER(Ffv1Config)]),
    "rcgpu_ffv1_stream_parse": (C.c_int, [_U8P, _SZ, _U8P, _SZ, C.POINTER(_VP)]),
    "rcgpu_ffv1_stream_free": (None, [_VP]),
    "rcgpu_ffv1_stream_get_info": (C.c_int, [_VP, C.POINTER(Ffv1StreamInfo)]),
    "rcgpu_ffv1_stream_get_tables": (C.c_int, [_VP, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rcgpu_ffv1_decoder_create_for_stream": (C.c_int, [C.POINTER(Ffv1Config), _VP, C.POINTER(_VP)]),
    "rcgpu_ffv1_decoder_last_kernel_times": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "rcgpu_compare_device": (C.c_int, [_VP, _VP, C.c_uint64, C.POINTER(C.c_uint64), _VP]),
    "rcgpu_compare_device_batch": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _VP, _VP]),
    "rcgpu_md5_device": (C.c_int, [C.POINTER(_VP), C.POINTER(C.c_uint64), C.c_uint32, _VP, _VP]),
    "rcgpu_flac_create": (C.c_int, [C.POINTER(FlacConfig), C.POINTER(_VP)]),
    "rcgpu_flac_destroy": (None, [_VP]),
    "rcgpu_flac_encode_host": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _SZ, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]),
    "rcgpu_flac_codec_private": (_SZ, [_VP, _VP, _SZ]),
    "rcgpu_mkv_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_VP)]),
    "rcgpu_mkv_add_video": (C.c_int, [_VP, _VP, _SZ, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rcgpu_mkv_add_audio": (C.c_int, [_VP, _VP, _SZ, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rcgpu_mkv_add_audio_pcm": (C.c_int, [_VP, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rcgpu_mkv_add_attachment": (C.c_int, [_VP, C.c_char_p, C.c_char_p, _VP, _SZ]),
    "rcgpu_mkv_add_tag": (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_char_p]),
    "rcgpu_mkv_begin": (C.c_int, [_VP]),
    "rcgpu_mkv_write_block": (C.c_int, [_VP, C.c_int, C.c_uint64, _VP, _SZ, C.c_int]),
    "rcgpu_mkv_update_codec_private": (C.c_int, [_VP, C.c_int, _VP, _SZ]),
    "rcgpu_mkv_close": (C.c_int, [_VP]),
    "rcgpu_md5": (None, [_VP, _SZ, _VP]),
    "rcgpu_crc32_ffv1": (C.c_uint32, [_VP, _SZ]),
    "rcgpu_dpx_padding_scan_device": (C.c_int, [C.POINTER(_VP), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), _VP]),
}

_lib = None


def lib() -> C.CDLL:
    """Load librcgpu.so (built in-tree by `make -C rawcooked_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RcgpuError(f"{LIB_PATH} is missing -- run __graft_entry__.build() (hipcc, gfx950); there is no fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        dbg = L.rcgpu_ffv1_debug_fetch
        dbg.restype, dbg.argtypes = C.c_longlong, [_VP, C.c_int, C.c_uint32, _VP, _SZ]
        L.rcgpu_ffv1_decoder_debug_window.restype, L.rcgpu_ffv1_decoder_debug_window.argtypes = C.c_int, [_VP, C.c_uint32]
        L.rcgpu_ffv1_decoder_debug_careful.restype, L.rcgpu_ffv1_decoder_debug_careful.argtypes = C.c_longlong, [_VP]
        # the library's per-device hash streams are CU-masked; they are given back while the HIP runtime is certainly still there, not in the
        # middle of the interpreter's and the runtime's teardown (profiles/r05_rocprof_cumask.txt: what a profiler makes of the other order)
        import atexit
        atexit.register(_at_exit)
        L.rcgpu_ffv1_decoder_debug_states_offset.restype, L.rcgpu_ffv1_decoder_debug_states_offset.argtypes = C.c_int, [_VP, C.c_uint64]
        _lib = L
    return _lib


_live_decoders = None


def _at_exit():
    """Before the interpreter and the HIP runtime go: decoders still alive are closed (one may have a batch decoded AHEAD on a thread of the
    library's -- a thread inside the runtime while it is torn down is a segmentation fault at exit, as route C's damage soak showed for the
    linked binary), then the library's CU-masked streams are given back."""
    for d in list(_live_decoders or ()):
        try:
            d.close()
        except Exception:
            pass
    if _lib is not None:
        _lib.rcgpu_release_device_streams()


def last_error() -> str:
    return lib().rcgpu_last_error().decode(errors="replace")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        e = (RcgpuUnsupported if rc == RcgpuUnsupported.code else RcgpuError)(f"{what} failed ({rc}): {last_error()}")
        e.code = rc
        raise e


def dpx_probe(data: bytes) -> ImageInfo:
    info = ImageInfo()
    _check(lib().rcgpu_dpx_probe(data, len(data), C.byref(info)), "rcgpu_dpx_probe")
    return info


def exr_probe(data: bytes) -> ImageInfo:
    info = ImageInfo()
    _check(lib().rcgpu_exr_probe(data, len(data), C.byref(info)), "rcgpu_exr_probe")
    return info


def md5_host_batch(bufs: list[bytes], device: int = 0) -> list[bytes]:
    n = len(bufs)
    keep = [C.create_string_buffer(b, len(b)) for b in bufs]
    ptrs = (_VP * n)(*[C.cast(k, _VP) for k in keep])
    sz = (C.c_uint64 * n)(*[len(b) for b in bufs])
    out = C.create_string_buffer(16 * n)
    _check(lib().rcgpu_md5_host_batch(ptrs, sz, n, out, device), "rcgpu_md5_host_batch")
    raw = out.raw
    return [raw[16 * i:16 * i + 16] for i in range(n)]


def analysis_host_batch(files: list[bytes], device: int = 0) -> list[tuple[bytes, bool, int]]:
    """Per file (MD5, scanned, first non-zero padding offset relative to the payload or -1): Input_Base.cpp:54-81 + DPX.cpp:501-608."""
    n = len(files)
    keep = [C.create_string_buffer(b, len(b)) for b in files]
    ptrs = (_VP * n)(*[C.cast(k, _VP) for k in keep])
    sz = (C.c_uint64 * n)(*[len(b) for b in files])
    out = C.create_string_buffer(16 * n)
    first = (C.c_uint64 * n)()
    scanned = C.create_string_buffer(n)
    _check(lib().rcgpu_analysis_host_batch(ptrs, sz, n, out, first, scanned, device), "rcgpu_analysis_host_batch")
    raw = out.raw
    return [(raw[16 * i:16 * i + 16], scanned.raw[i] != 0, -1 if first[i] == 0xFFFFFFFFFFFFFFFF else first[i]) for i in range(n)]


def config_from_record(record: bytes, width: int, height: int, pixfmt: int, line_bytes: int, flags: int = 0, context: int = 1) -> Ffv1Config:
    """Decoder configuration for a stream known only by its CodecPrivate (plus what the files' flavor says)."""
    cfg = Ffv1Config(width, height, pixfmt, line_bytes, 0, 0, 0, context, 1, 0, 0, flags, 0, 0)
    _check(lib().rcgpu_ffv1_config_from_record(record, len(record), C.byref(cfg)), "rcgpu_ffv1_config_from_record")
    return cfg


def tiff_probe(data: bytes) -> ImageInfo:
    info = ImageInfo()
    _check(lib().rcgpu_tiff_probe(data, len(data), C.byref(info)), "rcgpu_tiff_probe")
    return info


def wav_probe(data: bytes) -> AudioInfo:
    info = AudioInfo()
    _check(lib().rcgpu_wav_probe(data, len(data), C.byref(info)), "rcgpu_wav_probe")
    return info


def slices_to_grid(n: int) -> tuple[int, int]:
    h, v = C.c_uint32(), C.c_uint32()
    _check(lib().rcgpu_slices_to_grid(n, C.byref(h), C.byref(v)), "rcgpu_slices_to_grid")
    return h.value, v.value


def md5(data: bytes) -> bytes:
    out = C.create_string_buffer(16)
    lib().rcgpu_md5(data, len(data), out)
    return out.raw


class Ffv1Encoder:
    """Device FFV1 encoder (rcgpu_ffv1_*).  Raises when no HIP device is visible."""

    def __init__(self, width, height, pixfmt, line_bytes, num_h, num_v, slicecrc=1, context=1, max_batch=1, device=0, segments=0, flags=0, coder=1, level=3,
                 rc_span=0, slice_buffer_div=0):
        self.cfg = Ffv1Config(width, height, pixfmt, line_bytes, num_h, num_v, slicecrc, context, max_batch, device, segments, flags, coder, level,
                              rc_span, slice_buffer_div)
        self.h = _VP()
        _check(lib().rcgpu_ffv1_create(C.byref(self.cfg), C.byref(self.h)), "rcgpu_ffv1_create")
        self.max_packet = lib().rcgpu_ffv1_max_packet_bytes(self.h)

    def close(self):
        if self.h:
            lib().rcgpu_ffv1_destroy(self.h)
            self.h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:       # at interpreter shutdown the module's globals may be gone already
            pass

    def config_record(self) -> bytes:
        buf = C.create_string_buffer(8192)
        n = lib().rcgpu_ffv1_config_record(self.h, buf, 8192)
        return buf.raw[:n]

    def encode_host(self, payloads: list[bytes]) -> list[bytes]:
        n = len(payloads)
        keep = [C.create_string_buffer(p, len(p)) for p in payloads]
        ptrs = (_VP * n)(*[C.cast(k, _VP) for k in keep])
        outs = [C.create_string_buffer(self.max_packet) for _ in range(n)]
        optrs = (_VP * n)(*[C.cast(o, _VP) for o in outs])
        sizes = (_SZ * n)()
        _check(lib().rcgpu_ffv1_encode_host(self.h, ptrs, n, optrs, sizes), "rcgpu_ffv1_encode_host")
        return [C.string_at(outs[i], sizes[i]) for i in range(n)]      # only the packet, not the whole worst-case buffer

    def framemd5_last(self, n: int) -> tuple[list[bytes], int]:
        """MD5 of the last batch's first n frames as FFmpeg's rawvideo bytes, and the size of one such frame (`-f framemd5`)."""
        out = C.create_string_buffer(16 * n)
        fb = C.c_uint64()
        _check(lib().rcgpu_ffv1_framemd5_last(self.h, n, out, C.byref(fb)), "rcgpu_ffv1_framemd5_last")
        return [out.raw[16 * i:16 * i + 16] for i in range(n)], fb.value

    def encode_device(self, frame_ptrs: list[int], d_packets: int, packet_stride: int, d_sizes: int, stream: int = 0) -> None:
        n = len(frame_ptrs)
        ptrs = (_VP * n)(*frame_ptrs)
        _check(lib().rcgpu_ffv1_encode_device(self.h, ptrs, n, d_packets, packet_stride, d_sizes, stream), "rcgpu_ffv1_encode_device")

    def set_run_on(self, on: bool = True) -> None:
        """Run-on mode (rcgpu.h): the next batch is modelled and started while this one is in flight; a call joins the batch BEFORE it."""
        _check(lib().rcgpu_ffv1_set_run_on(self.h, 1 if on else 0), "rcgpu_ffv1_set_run_on")

    def join(self, stream: int = 0) -> None:
        """Makes `stream` wait for every batch issued so far."""
        _check(lib().rcgpu_ffv1_join(self.h, stream), "rcgpu_ffv1_join")

    def error_flags(self) -> int:
        """Device error word of the last batch (0 = fine); raises with the library's text when it is not."""
        f = C.c_uint32()
        _check(lib().rcgpu_ffv1_last_error_flags(self.h, C.byref(f)), "rcgpu_ffv1_last_error_flags")
        return f.value

    def batch_intervals(self, n: int = 63) -> list[float]:
        """ms from the end of one batch to the end of the next, for the last n pairs of batches (oldest first)."""
        buf = (C.c_float * max(1, n))()
        k = lib().rcgpu_ffv1_batch_intervals(self.h, buf, n)
        return [float(buf[i]) for i in range(k)]

    def kernel_times(self) -> dict[str, float]:
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = lib().rcgpu_ffv1_last_kernel_times(self.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    def kernel_launches(self) -> dict[str, int]:
        names = ["k_unpack", "k_model", "k_resolve", "k_rangecode", "k_footer", "k_scan", "k_gather", "k_rc_range", "k_rc_tails"]
        return {n: lib().rcgpu_ffv1_last_kernel_launches(self.h, i) for i, n in enumerate(names)}

    def stats(self) -> tuple[int, int]:
        d, b = C.c_uint64(), C.c_uint64()
        lib().rcgpu_ffv1_last_stats(self.h, C.byref(d), C.byref(b))
        return d.value, b.value

    def debug_fetch(self, what: int, chain: int, cap: int) -> bytes:
        buf = C.create_string_buffer(cap)
        n = lib().rcgpu_ffv1_debug_fetch(self.h, what, chain, buf, cap)
        if n < 0:
            raise RcgpuError(f"rcgpu_ffv1_debug_fetch({what}) -> {n}")
        return buf.raw[:n]


def sequence_plan(n_frames: int, batch: int, lanes: int) -> tuple[list[int], list[int]]:
    """rcgpu_sequence_plan: (lane of every frame, batch of every frame) -- the sharding rcgpu_ffv1_encode_sequence follows; needs no device."""
    lane = (C.c_uint32 * max(1, n_frames))()
    bat = (C.c_uint32 * max(1, n_frames))()
    _check(lib().rcgpu_sequence_plan(n_frames, batch, lanes, lane, bat), "rcgpu_sequence_plan")
    return list(lane[:n_frames]), list(bat[:n_frames])


def sequence_plan_lanes(n_frames: int, devices: int, lanes_per_device: int = 1) -> int:
    """rcgpu_sequence_plan_lanes: the lanes a job of n_frames really gets (a device per 8 frames at most) -- the pipeline's own decision."""
    return int(lib().rcgpu_sequence_plan_lanes(n_frames, devices, lanes_per_device))


def encode_sequence(cfg: Ffv1Config, n_frames: int, read_frame, packet_done, place_packet=None, batch=0, readers=0, writers=0,
                    in_ring_frames=0, out_ring_bytes=0, device_first=0, device_count=0, lanes_per_device=0, copy_streams=0, device_aliases=0, numa=0):
    """rcgpu_ffv1_encode_sequence: `read_frame(frame, dst_address, nbytes) -> int` fills a pinned upload slot, `packet_done(frame,
    address, size) -> int` receives each packet (writer threads), `place_packet(frame, size) -> address or None` is optional.
    Returns (SequenceStats, configuration record)."""
    rf = READ_FRAME_FN(lambda user, frame, dst, n: int(read_frame(frame, dst, n) or 0))
    pd = PACKET_DONE_FN(lambda user, frame, data, n: int(packet_done(frame, data, n) or 0))
    pp = PLACE_PACKET_FN(lambda user, frame, n: place_packet(frame, n) or 0) if place_packet else PLACE_PACKET_FN()
    io = SequenceIo(C.sizeof(SequenceIo), rf, pp, pd, None)
    opt = SequenceOptions(C.sizeof(SequenceOptions), device_first, device_count, batch, readers, writers, in_ring_frames, out_ring_bytes, lanes_per_device, copy_streams, device_aliases, 0, 0, numa)
    st = SequenceStats()
    rec = C.create_string_buffer(8192)
    rs = _SZ(8192)
    _check(lib().rcgpu_ffv1_encode_sequence(C.byref(cfg), n_frames, C.byref(io), C.byref(opt), C.byref(st), rec, C.byref(rs)), "rcgpu_ffv1_encode_sequence")
    return st, rec.raw[:rs.value]


def encode_sequence_memory(cfg: Ffv1Config, frame_addrs: list[int], n_frames: int, out_addrs: list[int], out_cap: int, batch=0, readers=0, writers=0,
                           device_first=0, device_count=0, lanes_per_device=0, in_ring_frames=0, copy_streams=0, device_aliases=0, numa=0, frames_pinned=0, run_on=0):
    """rcgpu_ffv1_encode_sequence_memory: frame i = frame_addrs[i % len], packet i -> out_addrs[i % len].  Returns (stats, sizes)."""
    fin = (_VP * len(frame_addrs))(*frame_addrs)
    fout = (_VP * len(out_addrs))(*out_addrs) if out_addrs else None
    sizes = (C.c_uint64 * n_frames)()
    opt = SequenceOptions(C.sizeof(SequenceOptions), device_first, device_count, batch, readers, writers, in_ring_frames, 0, lanes_per_device, copy_streams, device_aliases, frames_pinned, run_on, numa)
    st = SequenceStats()
    _check(lib().rcgpu_ffv1_encode_sequence_memory(C.byref(cfg), fin, len(frame_addrs), n_frames, fout, len(out_addrs), out_cap, sizes, C.byref(opt), C.byref(st), None, None),
           "rcgpu_ffv1_encode_sequence_memory")
    return st, list(sizes)


class KeptFile(C.Structure):
    """rcgpu_kept_file (include/rcgpu.h)."""
    _fields_ = [("slot", C.c_uint32), ("flags", C.c_uint32), ("before", C.c_void_p), ("before_size", C.c_uint64),
                ("after", C.c_void_p), ("after_size", C.c_uint64), ("on_disk", C.c_void_p), ("on_disk_size", C.c_uint64), ("on_disk_path", C.c_char_p)]


class KeptVerdict(C.Structure):
    """rcgpu_kept_verdict (include/rcgpu.h)."""
    _fields_ = [("md5", C.c_uint8 * 16), ("first_diff", C.c_uint64)]


class Ffv1Stream:
    """rcgpu_ffv1_stream: a stream as parameters::Parse reads it (FFV1_Parameters.cpp:23-183) -- record = CodecPrivate or b"" (version 0 / 1:
    the header is inside `packet`), packet = the first frame of the track."""

    def __init__(self, record: bytes, packet: bytes):
        self.h = _VP()
        _check(lib().rcgpu_ffv1_stream_parse(record if record else None, len(record), packet, len(packet), C.byref(self.h)), "rcgpu_ffv1_stream_parse")

    def info(self) -> Ffv1StreamInfo:
        i = Ffv1StreamInfo()
        _check(lib().rcgpu_ffv1_stream_get_info(self.h, C.byref(i)), "rcgpu_ffv1_stream_get_info")
        return i

    def tables(self, set_index: int):
        """(state transitions: 256 bytes, quantisation tables: int16 [5][256], coded initial states: bytes, b"" when the set starts at 128)."""
        import numpy as np
        one = (C.c_uint8 * 256)(); q = np.zeros((5, 256), dtype=np.int16); n = C.c_uint64(0)
        _check(lib().rcgpu_ffv1_stream_get_tables(self.h, set_index, one, q.ctypes.data, None, 0, C.byref(n)), "rcgpu_ffv1_stream_get_tables")
        init = (C.c_uint8 * max(1, n.value))()
        if n.value:
            _check(lib().rcgpu_ffv1_stream_get_tables(self.h, set_index, None, None, init, n.value, None), "rcgpu_ffv1_stream_get_tables")
        return bytes(one), q, bytes(init)[:n.value]

    def close(self):
        if self.h:
            lib().rcgpu_ffv1_stream_free(self.h)
            self.h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:       # at interpreter shutdown the module's globals may be gone already
            pass


class Ffv1Decoder:
    """Device FFV1 decoder + pack (the --check half).  Buffers are device pointers (ints)."""

    def __init__(self, width, height, pixfmt, line_bytes, num_h=0, num_v=0, slicecrc=1, context=1, max_batch=1, device=0, flags=0, coder=1, level=3, stream=None):
        """stream = None: the stream this library's encoder writes for these fields; an Ffv1Stream: whatever it says (num_h ... level ignored)"""
        self.cfg = Ffv1Config(width, height, pixfmt, line_bytes, num_h, num_v, slicecrc, context, max_batch, device, 0, flags, coder, level)
        self.h = _VP()
        if stream is not None:
            _check(lib().rcgpu_ffv1_decoder_create_for_stream(C.byref(self.cfg), stream.h, C.byref(self.h)), "rcgpu_ffv1_decoder_create_for_stream")
        else:
            _check(lib().rcgpu_ffv1_decoder_create(C.byref(self.cfg), C.byref(self.h)), "rcgpu_ffv1_decoder_create")
        global _live_decoders
        if _live_decoders is None:
            import weakref
            _live_decoders = weakref.WeakSet()
        _live_decoders.add(self)

    def close(self):
        if self.h:
            lib().rcgpu_ffv1_decoder_destroy(self.h)
            self.h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:       # at interpreter shutdown the module's globals may be gone already
            pass

    def debug_window(self, nbytes: int) -> None:
        """Test hook: the sample decoder's byte window is filled to `nbytes` (1..7) instead of 7, so that samples outrun it and go through the careful path."""
        if lib().rcgpu_ffv1_decoder_debug_window(self.h, nbytes) != 0:
            raise RcgpuError("rcgpu_ffv1_decoder_debug_window failed")

    def debug_states_offset(self, nbytes: int) -> None:
        """Test hook: the context-state arrays begin `nbytes` (multiple of 256, <= 64 MiB) into their allocation from the next batch on."""
        if lib().rcgpu_ffv1_decoder_debug_states_offset(self.h, nbytes) != 0:
            raise RcgpuError("rcgpu_ffv1_decoder_debug_states_offset failed")

    def debug_careful(self) -> int:
        """Samples (per wavefront) of the last batch that were decoded a second time, carefully."""
        return int(lib().rcgpu_ffv1_decoder_debug_careful(self.h))

    def decode_device(self, packet_ptrs: list[int], packet_sizes: list[int], payload_ptrs: list[int], stream: int = 0, check: bool = True) -> int:
        n = len(packet_ptrs)
        pk = (_VP * n)(*packet_ptrs)
        sz = (C.c_uint64 * n)(*packet_sizes)
        out = (_VP * n)(*payload_ptrs)
        flags = C.c_uint32(0)
        _check(lib().rcgpu_ffv1_decoder_decode_device(self.h, pk, sz, n, out, C.byref(flags) if check else None, stream), "rcgpu_ffv1_decoder_decode_device")
        return flags.value

    def decode_host(self, packets: list[bytes], payload_bytes: int) -> list[bytes]:
        n = len(packets)
        keep = [C.create_string_buffer(p, len(p)) for p in packets]
        pk = (_VP * n)(*[C.cast(k, _VP) for k in keep])
        sz = (C.c_uint64 * n)(*[len(p) for p in packets])
        outs = [C.create_string_buffer(payload_bytes) for _ in range(n)]
        op = (_VP * n)(*[C.cast(o, _VP) for o in outs])
        _check(lib().rcgpu_ffv1_decoder_decode_host(self.h, pk, sz, n, op), "rcgpu_ffv1_decoder_decode_host")
        return [o.raw for o in outs]

    class PacketBatch:
        """Packets at addresses that stay put (what a mapped Matroska file gives the reference): for decode_keep_hint + decode_keep."""
        def __init__(self, packets: list[bytes]):
            self.n = len(packets)
            self.keep = [C.create_string_buffer(p, len(p)) for p in packets]
            self.pk = (_VP * self.n)(*[C.cast(k, _VP) for k in self.keep])
            self.sz = (C.c_uint64 * self.n)(*[len(p) for p in packets])

    def decode_keep(self, packets) -> None:
        """Decodes the packets (a list of bytes or a PacketBatch); payload i stays on the device as slot i (rcgpu_ffv1_decoder_decode_keep)."""
        b = packets if isinstance(packets, Ffv1Decoder.PacketBatch) else Ffv1Decoder.PacketBatch(packets)
        _check(lib().rcgpu_ffv1_decoder_decode_keep(self.h, b.pk, b.sz, b.n), "rcgpu_ffv1_decoder_decode_keep")

    def decode_keep_hint(self, batch) -> None:
        """Starts decoding `batch` (a PacketBatch) ahead of the decode_keep that will ask for it (rcgpu_ffv1_decoder_decode_keep_hint)."""
        _check(lib().rcgpu_ffv1_decoder_decode_keep_hint(self.h, batch.pk, batch.sz, batch.n), "rcgpu_ffv1_decoder_decode_keep_hint")

    def decode_keep_hint_file(self, path: str, offsets: list[int], sizes: list[int]) -> None:
        """Starts decoding the packets at `offsets` of the file `path` ahead (rcgpu_ffv1_decoder_decode_keep_hint_file)."""
        n = len(offsets)
        _check(lib().rcgpu_ffv1_decoder_decode_keep_hint_file(self.h, path.encode(), (C.c_uint64 * n)(*offsets), (C.c_uint64 * n)(*sizes), n), "rcgpu_ffv1_decoder_decode_keep_hint_file")

    def decode_keep_adopt(self) -> None:
        """The batch decoded ahead becomes the current one (rcgpu_ffv1_decoder_decode_keep_adopt)."""
        _check(lib().rcgpu_ffv1_decoder_decode_keep_adopt(self.h), "rcgpu_ffv1_decoder_decode_keep_adopt")

    def decode_keep_fd(self, fd: int, offsets: list[int], sizes: list[int]) -> None:
        """The same with the packets at `offsets` of the open file `fd` (rcgpu_ffv1_decoder_decode_keep_fd)."""
        n = len(offsets)
        _check(lib().rcgpu_ffv1_decoder_decode_keep_fd(self.h, fd, (C.c_uint64 * n)(*offsets), (C.c_uint64 * n)(*sizes), n), "rcgpu_ffv1_decoder_decode_keep_fd")

    def kept_to_host(self, slot: int, payload_bytes: int) -> bytes:
        out = C.create_string_buffer(payload_bytes)
        _check(lib().rcgpu_ffv1_decoder_kept_to_host(self.h, slot, out), "rcgpu_ffv1_decoder_kept_to_host")
        return out.raw

    def verify_kept(self, files: list[dict], begin_only: bool = False) -> list[tuple[bytes, int]]:
        """files: dicts with slot, before, after (bytes), on_disk (bytes or None), md5 (bool) -> per file (md5, first differing offset or -1):
        frame_writer's CheckMD5 and CheckFile (FileWriter.cpp:464-727) for a batch of rebuilt files, on the device."""
        n = len(files)
        arr = (KeptFile * n)()
        keep = []
        for i, f in enumerate(files):
            arr[i].slot = f["slot"]
            arr[i].flags = 1 if f.get("md5", True) else 0
            if f.get("on_disk_path"):
                arr[i].on_disk_path = os.fsencode(f["on_disk_path"])
            for name in ("before", "after", "on_disk"):
                v = f.get(name)
                if v is None:
                    continue
                b = C.create_string_buffer(bytes(v), len(v)) if len(v) else C.create_string_buffer(1)
                keep.append(b)
                setattr(arr[i], name, C.cast(b, _VP))
                setattr(arr[i], name + "_size", len(v))
        if begin_only:
            _check(lib().rcgpu_ffv1_decoder_verify_kept_begin(self.h, arr, n), "rcgpu_ffv1_decoder_verify_kept_begin")
            self._begun = n
            return []
        out = (KeptVerdict * n)()
        _check(lib().rcgpu_ffv1_decoder_verify_kept(self.h, arr, n, out), "rcgpu_ffv1_decoder_verify_kept")
        return [(bytes(bytearray(o.md5)), -1 if o.first_diff == 0xFFFFFFFFFFFFFFFF else o.first_diff) for o in out]

    def verify_kept_end(self) -> list[tuple[bytes, int]]:
        """The verdicts of the verification begun with verify_kept(..., begin_only=True)."""
        n = getattr(self, "_begun", 0) or 1
        out = (KeptVerdict * n)()
        _check(lib().rcgpu_ffv1_decoder_verify_kept_end(self.h, out), "rcgpu_ffv1_decoder_verify_kept_end")
        self._begun = 0
        return [(bytes(bytearray(o.md5)), -1 if o.first_diff == 0xFFFFFFFFFFFFFFFF else o.first_diff) for o in out]

    def batch_intervals(self, n: int = 63) -> list[float]:
        """ms from the end of one batch to the end of the next, for the last n pairs of batches (oldest first)."""
        buf = (C.c_float * max(1, n))()
        k = lib().rcgpu_ffv1_batch_intervals(self.h, buf, n)
        return [float(buf[i]) for i in range(k)]

    def kernel_times(self) -> dict[str, float]:
        ms = (C.c_float * 3)()
        lib().rcgpu_ffv1_decoder_last_kernel_times(self.h, ms)
        return {"k_dec_split+k_dec_crc": float(ms[0]), "k_dec_slices": float(ms[1]), "k_pack": float(ms[2])}


def compare_device(a: int, b: int, n: int, stream: int = 0) -> int:
    """-> index of the first differing byte, or -1."""
    r = C.c_uint64()
    _check(lib().rcgpu_compare_device(a, b, n, C.byref(r), stream), "rcgpu_compare_device")
    return -1 if r.value == 0xFFFFFFFFFFFFFFFF else r.value


def compare_device_batch(a: list[int], b: list[int], sizes: list[int], stream: int = 0) -> list[int]:
    """-> per pair the index of the first differing byte, or -1."""
    n = len(a)
    out = (C.c_uint64 * n)()
    _check(lib().rcgpu_compare_device_batch((_VP * n)(*a), (_VP * n)(*b), (C.c_uint64 * n)(*sizes), n, out, stream), "rcgpu_compare_device_batch")
    return [-1 if v == 0xFFFFFFFFFFFFFFFF else v for v in out]


def dpx_padding_scan_device(ptrs: list[int], pixfmt: int, width: int, height: int, flags: int = 0, stream: int = 0) -> list[int]:
    """-> per payload the offset of the first non-zero padding position (DPX.cpp:501-608), or 2**64 - 1."""
    n = len(ptrs)
    out = (C.c_uint64 * n)()
    _check(lib().rcgpu_dpx_padding_scan_device((_VP * n)(*ptrs), n, pixfmt, width, height, flags, out, stream), "rcgpu_dpx_padding_scan_device")
    return list(out)


def md5_device(ptrs: list[int], sizes: list[int], stream: int = 0) -> list[bytes]:
    n = len(ptrs)
    out = C.create_string_buffer(16 * n)
    _check(lib().rcgpu_md5_device((_VP * n)(*ptrs), (C.c_uint64 * n)(*sizes), n, out, stream), "rcgpu_md5_device")
    raw = out.raw
    return [raw[16 * i:16 * i + 16] for i in range(n)]


class FlacEncoder:
    def __init__(self, channels, sample_rate, bits_per_sample, block_size=0, max_lpc_order=8, device=0):
        self.cfg = FlacConfig(channels, sample_rate, bits_per_sample, block_size, max_lpc_order, device)
        self.h = _VP()
        _check(lib().rcgpu_flac_create(C.byref(self.cfg), C.byref(self.h)), "rcgpu_flac_create")

    def close(self):
        if self.h:
            lib().rcgpu_flac_destroy(self.h)
            self.h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:       # at interpreter shutdown the module's globals may be gone already
            pass

    def encode(self, pcm: bytes) -> tuple[list[bytes], bytes]:
        """Returns (FLAC frames, CodecPrivate = 'fLaC' + STREAMINFO)."""
        cap = len(pcm) + len(pcm) // 4 + 65536
        out = C.create_string_buffer(cap)
        fcap = len(pcm) // 64 + 64
        sizes = (C.c_uint32 * fcap)()
        n = C.c_uint32()
        keep = C.create_string_buffer(pcm, len(pcm))
        _check(lib().rcgpu_flac_encode_host(self.h, keep, len(pcm), out, cap, sizes, fcap, C.byref(n)), "rcgpu_flac_encode_host")
        frames, off = [], 0
        raw = out.raw                      # one copy, not one per frame
        for i in range(n.value):
            frames.append(raw[off:off + sizes[i]])
            off += sizes[i]
        cp = C.create_string_buffer(256)
        k = lib().rcgpu_flac_codec_private(self.h, cp, 256)
        return frames, cp.raw[:k]


class MkvMuxer:
    def __init__(self, path: str, overwrite: bool = True):
        self.h = _VP()
        _check(lib().rcgpu_mkv_open(path.encode(), 1 if overwrite else 0, C.byref(self.h)), "rcgpu_mkv_open")

    def add_video(self, codec_private: bytes, width, height, fps_num=24, fps_den=1) -> int:
        t = lib().rcgpu_mkv_add_video(self.h, codec_private, len(codec_private), width, height, fps_num, fps_den)
        if t < 0:
            raise RcgpuError("rcgpu_mkv_add_video: " + last_error())
        return t

    def add_audio(self, codec_private: bytes, channels, sample_rate, bits) -> int:
        t = lib().rcgpu_mkv_add_audio(self.h, codec_private, len(codec_private), channels, sample_rate, bits)
        if t < 0:
            raise RcgpuError("rcgpu_mkv_add_audio: " + last_error())
        return t

    def add_attachment(self, name: str, data: bytes, mime: str = "application/octet-stream"):
        _check(lib().rcgpu_mkv_add_attachment(self.h, name.encode(), mime.encode(), data, len(data)), "rcgpu_mkv_add_attachment")

    def add_tag(self, track, name, value):
        _check(lib().rcgpu_mkv_add_tag(self.h, track, name.encode(), value.encode()), "rcgpu_mkv_add_tag")

    def begin(self):
        _check(lib().rcgpu_mkv_begin(self.h), "rcgpu_mkv_begin")

    def write_block(self, track: int, pts_ns: int, data: bytes, keyframe: bool = True):
        _check(lib().rcgpu_mkv_write_block(self.h, track, pts_ns, data, len(data), 1 if keyframe else 0), "rcgpu_mkv_write_block")

    def close(self):
        if self.h:
            h, self.h = self.h, _VP()
            _check(lib().rcgpu_mkv_close(h), "rcgpu_mkv_close")


# --- the reference's job vocabulary (Source/CLI/Output.h:21-53) -------------------------------------------------
@dataclass
class Stream:
    """`stream` of Output.h:21-33."""
    FileName: str = ""
    FileName_Template: str = ""
    FileName_StartNumber: str = ""
    FileList: str = ""
    Flavor: str = ""
    Slices: str = ""
    FrameRate: str = ""


@dataclass
class Attachment:
    """`attachment` of Output.h:34-38."""
    FileName_In: str = ""
    FileName_Out: str = ""


@dataclass
class Output:
    """`output` of Output.h:41-53: fill Streams/Attachments, call Process()."""
    Streams: list = field(default_factory=list)
    Attachments: list = field(default_factory=list)

    def Process(self, OutputFileName: str, rawcooked_reversibility_FileName: str | None = None, OutputOptions: dict | None = None,
                device_first: int = 0, device_count: int = 0) -> int:
        opts = {"c:v": "ffv1", "c:a": "flac", "coder": "1", "context": "1", "g": "1", "level": "3", "slicecrc": "1", "y": ""}   # Global.cpp:938-989
        opts.update(OutputOptions or {})
        keep = []

        def s(x):
            if not x:
                return None
            b = x.encode()
            keep.append(b)
            return b

        streams = (_Stream * len(self.Streams))()
        for i, st in enumerate(self.Streams):
            streams[i] = _Stream(s(st.FileName_Template or st.FileName), s(st.FileName_StartNumber), s(st.FileList), s(st.Flavor),
                                 s(st.FrameRate), int(st.Slices) if st.Slices else (0 if st.Flavor.startswith("WAV/") else 0xFFFFFFFF), 0)
        atts = (_Attachment * max(1, len(self.Attachments)))()
        for i, a in enumerate(self.Attachments):
            atts[i] = _Attachment(s(a.FileName_In), s(a.FileName_Out))
        flat = []
        for k, v in sorted(opts.items()):
            flat += [k.encode(), str(v).encode()]
        arr = (C.c_char_p * len(flat))(*flat)
        job = _Job(streams, len(self.Streams), atts, len(self.Attachments), s(rawcooked_reversibility_FileName), s(OutputFileName), None,
                   arr, len(flat), device_first, device_count)
        return lib().rcgpu_encode(C.byref(job))
