"""GPU: the BASELINE config shapes at full size, through size-independent properties (the oracle needs seconds per 4K frame):
encode -> device decode -> device compare/MD5 round trip, and one frame of each checked by the oracle's decoder."""
import hashlib

import pytest
import torch

import oracle_binding as ob
from rawcooked_amd import api, synth

pytestmark = pytest.mark.gpu


def roundtrip(w, h, pixfmt, slices, n, kind, oracle_frames=1, line_pad=True):
    bits, nc, bpp, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    srcs = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, kind, seed=200 + i), pixfmt, line_pad)
        srcs.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    dsrc = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in srcs]
    stride = (enc.max_packet + 255) & ~255
    dpk = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
    dsz = torch.zeros(n, dtype=torch.int64, device="cuda")
    enc.encode_device([t.data_ptr() for t in dsrc], dpk.data_ptr(), stride, dsz.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sizes = dsz.cpu().tolist()
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    dout = [torch.empty(len(srcs[0]), dtype=torch.uint8, device="cuda") for _ in range(n)]
    assert dec.decode_device([dpk.data_ptr() + i * stride for i in range(n)], sizes, [t.data_ptr() for t in dout]) == 0
    for i in range(n):
        assert api.compare_device(dout[i].data_ptr(), dsrc[i].data_ptr(), len(srcs[i])) == -1
    assert api.md5_device([dout[0].data_ptr()], [len(srcs[0])])[0] == hashlib.md5(srcs[0]).digest()
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    for i in range(oracle_frames):
        pk = bytes(dpk[i * stride:i * stride + sizes[i]].cpu().numpy())
        assert ob.decode_payload(p, pk, line_bytes) == srcs[i], "oracle decoder disagrees"
    enc.close(); dec.close()
    return sizes


def test_config1_2k_10bit_64_slices(built):
    roundtrip(2048, 1556, synth.PIX_RGB10_FILLEDA_BE, 64, 3, "film")


def test_config2_4k_16bit_64_slices(built):
    roundtrip(4096, 2160, synth.PIX_RGB16_BE, 64, 2, "film")


def test_config2_4k_16bit_576_slices_default(built):
    """the slice count the reference itself would pass for this picture (DPX.cpp:428-441): 576 = 32 x 18"""
    assert api.lib().rcgpu_reference_slices(4096, 2160, 16, 1) == 576
    roundtrip(4096, 2160, synth.PIX_RGB16_BE, 576, 2, "film")


def test_config4_8k_tiff_16bit_576_slices(built):
    roundtrip(8192, 4320, synth.PIX_RGB16_LE, 576, 1, "film", line_pad=False)


def test_flat_and_noise_extremes_4k(built):
    s_flat = roundtrip(4096, 2160, synth.PIX_RGB16_BE, 64, 1, "flat", oracle_frames=1)
    s_noise = roundtrip(4096, 2160, synth.PIX_RGB16_BE, 64, 1, "noise", oracle_frames=0)
    assert s_flat[0] < 4096 * 2160 * 6 // 50 and s_noise[0] > 4096 * 2160 * 6
