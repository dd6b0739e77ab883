"""GPU parity, stage by stage: every intermediate of the device pipeline against the oracle's trace of the same
slice (planes -> symbols -> decisions -> slice bytes -> packet).  Bit-exact: this is integer/byte work."""
import os
import time
import numpy as np
import pytest

import oracle_binding as ob
from rawcooked_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = [
    # w, h, pixfmt, slices, kind, frames
    (64, 48, synth.PIX_RGB16_BE, 4, "film", 2),
    (50, 38, synth.PIX_RGB10_FILLEDA_BE, 6, "film", 1),
    (130, 67, synth.PIX_RGB16_LE, 9, "noise", 3),
    (96, 64, synth.PIX_RGB16_BE, 4, "flat", 1),
    (257, 131, synth.PIX_RGB8, 16, "film", 2),
    (600, 200, synth.PIX_RGB16_BE, 1, "noise", 2),           # a slice large enough for its coded bytes to lie inside its symbol area (round 6)
]


# rc_span: 1 = one lane codes a whole slice, N = split coder with N-piece spans (0 = automatic: split when chains are few, as here)
@pytest.mark.parametrize("rc_span", [0, 1, 8])
@pytest.mark.parametrize("w,h,pixfmt,slices,kind,nframes", CASES)
def test_stages_match_oracle(built, w, h, pixfmt, slices, kind, nframes, rc_span):
    bits, nc, bpp, be = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    payloads = []
    for i in range(nframes):
        comp = synth.components(w, h, nc, bits, kind, seed=100 + i)
        pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
        payloads.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=nframes, segments=1, rc_span=rc_span)   # whole decision stream stays resident: every stage can be fetched
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    assert enc.config_record() == ob.config_record(p)
    packets = enc.encode_host(payloads)
    S = nh * nv
    gpu_planes = np.frombuffer(enc.debug_fetch(0, 0, nframes * nc * w * h * 4), dtype=np.int32).reshape(nframes, nc, h, w)
    gpu_sym = np.frombuffer(enc.debug_fetch(1, 0, nframes * nc * w * h * 4), dtype=np.uint32).reshape(nframes, -1)
    for f in range(nframes):
        planes = ob.unpack(p, payloads[f], line_bytes, nc)
        assert np.array_equal(gpu_planes[f], planes), f"frame {f}: unpack/RCT planes differ"
        sym_off = 0
        for sy in range(nv):
            for sx in range(nh):
                x0, x1 = sx * w // nh, (sx + 1) * w // nh
                y0, y1 = sy * h // nv, (sy + 1) * h // nv
                nsamp = (x1 - x0) * (y1 - y0) * nc
                sym, dec, raw = ob.trace_slice(p, planes, sx, sy, nsamp)
                chain = f * S + sy * nh + sx
                g = gpu_sym[f, sym_off:sym_off + nsamp]
                assert np.array_equal(g, sym), f"chain {chain}: symbols differ, first at {np.flatnonzero(g != sym)[:5]}"
                gdec = np.frombuffer(enc.debug_fetch(3, chain, len(dec) * 2 + 64), dtype=np.uint16)
                assert len(gdec) == len(dec), f"chain {chain}: {len(gdec)} decisions on GPU, {len(dec)} in oracle"
                assert np.array_equal(gdec, dec), f"chain {chain}: decisions differ, first at {np.flatnonzero(gdec != dec)[:5]}"
                graw = enc.debug_fetch(4, chain, len(raw) + 64)
                assert graw == raw[:-8], f"chain {chain}: range-coder bytes differ ({len(graw)} vs {len(raw) - 8})"
                sym_off += nsamp
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}: packet differs"
        assert ob.decode_payload(p, packets[f], line_bytes) == payloads[f]
    enc.close()


@pytest.mark.parametrize("rc_span", [1, 8, 13, 64, 100000])
@pytest.mark.parametrize("segments", [0, 2, 5, 16, 64])
def test_segmented_handover_is_bit_exact(built, segments, rc_span):
    """The k_resolve -> k_rangecode hand-over in windows (resumable kernels, two or three streams) must not change a byte, whether one
    lane codes a whole slice (rc_span 1) or the coder is split into spans of rc_span pieces (k_rc_range / k_rangecode<true> / k_rc_tails)."""
    w, h, pixfmt, nh, nv, nframes = 200, 120, synth.PIX_RGB16_BE, 3, 2, 3
    payloads = []
    for i in range(nframes):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film" if i else "noise", seed=50 + i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=nframes, segments=segments, rc_span=rc_span)
    for rep in range(2):          # second call reuses every buffer and resume record
        packets = enc.encode_host(payloads)
        for f in range(nframes):
            assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"segments={segments} rc_span={rc_span} call {rep} frame {f}"
    enc.close()


def test_many_frames_cross_sub_batches(built):
    """More frames than debug_fetch returns planes for (16) and more than one wavefront of chains."""
    w, h, pixfmt, nh, nv, nframes = 96, 64, synth.PIX_RGB10_FILLEDA_BE, 2, 2, 37
    payloads = []
    for i in range(nframes):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 10, "film", seed=i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=nframes)
    packets = enc.encode_host(payloads)
    for f in range(nframes):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
    enc.close()


@pytest.mark.parametrize("rc_span", [1, 8], ids=["whole-slice coder", "split coder"])
@pytest.mark.parametrize("segments", [1, 5, 32])
@pytest.mark.parametrize("w,h,pixfmt,slices,nframes", [(200, 120, synth.PIX_RGB16_BE, 6, 5), (96, 64, synth.PIX_RGB10_FILLEDA_BE, 4, 70), (512, 270, synth.PIX_RGB16_BE, 1, 3)])
def test_run_on_mode_is_bit_exact(built, w, h, pixfmt, slices, nframes, segments, rc_span):
    """Run-on mode (rcgpu_ffv1_set_run_on): batch k+1 is modelled and started while batch k is in flight, two banks of per-batch buffers,
    the windows' ring running through.  Seven batches of different pictures and sizes (the last ones short), each into buffers of its own:
    every packet equals the oracle's, whatever was in flight beside it; then back to one batch at a time with the same encoder.
    With either mapping of the range coder (round 6: the split coder -- k_rc_range's nine-instruction chain, spans, tails -- runs on as well:
    checkpoints per bank, k_resolve and the chain on streams of their own, the tail behind k_rc_tails)."""
    import torch
    bits, nc, bpp, be = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    sizes_of = [nframes, nframes, max(1, nframes // 2), nframes, 1, nframes, max(1, nframes - 1)]
    line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, "flat", seed=0), pixfmt, True)[1]
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=nframes, segments=segments, rc_span=rc_span)
    stride = (enc.max_packet + 255) & ~255
    st = torch.cuda.Stream()
    batches = []
    for b, n in enumerate(sizes_of):
        pls = [synth.pack_payload(synth.components(w, h, nc, bits, "film" if (b + i) % 3 else "noise", seed=1000 * b + i), pixfmt, True)[0] for i in range(n)]
        batches.append((pls, [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in pls],
                        torch.full((n * stride,), 0x5A, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int64, device="cuda")))
    torch.cuda.synchronize()

    def check(b):
        pls, _, d_pk, d_sz = batches[b]
        sz = d_sz.cpu().tolist()
        for i, pl in enumerate(pls):
            got = bytes(d_pk[i * stride:i * stride + sz[i]].cpu().numpy())
            assert got == ob.encode_payload(p, pl, line_bytes), f"batch {b} frame {i}"

    enc.set_run_on(True)
    for b, (pls, d_in, d_pk, d_sz) in enumerate(batches[:5]):
        enc.encode_device([t.data_ptr() for t in d_in], d_pk.data_ptr(), stride, d_sz.data_ptr(), st.cuda_stream)
        if b >= 1:                      # the call for batch b has joined batch b - 1 to the stream
            st.synchronize(); check(b - 1)
    enc.join(st.cuda_stream); st.synchronize(); check(4)
    assert enc.error_flags() == 0
    enc.set_run_on(False)
    for b in (5, 6):
        pls, d_in, d_pk, d_sz = batches[b]
        enc.encode_device([t.data_ptr() for t in d_in], d_pk.data_ptr(), stride, d_sz.data_ptr(), st.cuda_stream)
        st.synchronize(); check(b)
    enc.set_run_on(True)               # and on again: the banks and the ring carry on
    for b in (0, 1):
        pls, d_in, d_pk, d_sz = batches[b]
        d_pk.fill_(0)
        enc.encode_device([t.data_ptr() for t in d_in], d_pk.data_ptr(), stride, d_sz.data_ptr(), st.cuda_stream)
    enc.join(st.cuda_stream); st.synchronize(); check(0); check(1)
    enc.close()


def test_run_on_mode_and_the_host_conveniences(built):
    """rcgpu_ffv1_encode_host and rcgpu_ffv1_framemd5_last with a run-on encoder: encode_device leaves the batch unjoined in that mode (its
    coder, footer, scan and gather are still on their streams), so both join it before they read sizes and packets / before the rawvideo
    bytes go into the bank's symbol buffer.  Several batches back to back, packets against the oracle, frame sums against run-on off."""
    w, h, pixfmt, nh, nv, n = 200, 120, synth.PIX_RGB16_BE, 3, 2, 6
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    batches = []
    for b in range(4):
        pls = [synth.pack_payload(synth.components(w, h, nc, bits, "film" if (b + i) % 2 else "noise", seed=50 * b + i), pixfmt, True) for i in range(n - b)]
        batches.append([x[0] for x in pls])
        line_bytes = pls[0][1]
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n, rc_span=1)
    want_sums = []
    for pls in batches:
        enc.encode_host(pls)
        want_sums.append(enc.framemd5_last(len(pls)))
    enc.set_run_on(True)
    for b, pls in enumerate(batches):
        got = enc.encode_host(pls)
        for i, pl in enumerate(pls):
            assert got[i] == ob.encode_payload(p, pl, line_bytes), f"batch {b} frame {i}"
        assert enc.framemd5_last(len(pls)) == want_sums[b]
        assert enc.error_flags() == 0
    enc.close()


@pytest.mark.parametrize("pixfmt,w,h,slices", [(synth.PIX_RGB16_BE, 200, 120, 6), (synth.PIX_RGB10_FILLEDA_LE, 96, 64, 4), (synth.PIX_RGBA16_BE, 64, 48, 4),
                                                (synth.PIX_Y16_LE, 80, 40, 4), (synth.PIX_RGB8, 90, 50, 4)])
@pytest.mark.parametrize("segments", [1, 3])
def test_compact_context_model(built, pixfmt, w, h, slices, segments):
    """context = 2: 5-input model with compact level maps, adaptive states resident in LDS -- same bytes as the oracle, and the
    device decoder (states in HBM) rebuilds the payload."""
    import torch
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    payloads = []
    for i in range(3):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, ["film", "noise", "flat"][i], seed=70 + i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 2)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 2, max_batch=3, segments=segments)
    assert enc.config_record() == ob.config_record(p)
    packets = enc.encode_host(payloads)
    for f in range(3):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 2, max_batch=3)
    dpk = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in packets]
    dout = [torch.empty(len(x), dtype=torch.uint8, device="cuda") for x in payloads]
    assert dec.decode_device([t.data_ptr() for t in dpk], [len(x) for x in packets], [t.data_ptr() for t in dout]) == 0
    for f in range(3):
        assert bytes(dout[f].cpu().numpy()) == payloads[f]
    enc.close(); dec.close()


import json as _json
import os as _os
_G = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
_VEC = _json.load(open(_os.path.join(_G, "vectors.json")))


@pytest.mark.parametrize("v", _VEC["ffv1"], ids=lambda v: v["name"])
def test_encode_golden_payloads(built, v):
    """Every flavor: committed payload -> device encoder == the committed packet the REAL reference accepted and decoded
    byte-identically (tests/golden/make_golden.py).  This check needs neither the oracle nor /root/reference at run time."""
    payloads = [open(_os.path.join(_G, f["payload"]), "rb").read() for f in v["frames"]]
    packets = [open(_os.path.join(_G, f["packet"]), "rb").read() for f in v["frames"]]
    enc = api.Ffv1Encoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["num_h"], v["num_v"], 1, 1, max_batch=len(payloads), flags=v["flags"], coder=v["coder"])
    assert enc.config_record().hex() == v["config_record"]
    assert enc.encode_host(payloads) == packets
    enc.close()


FLAVORS = [  # pixfmt, layout flags -- the bit-packed DPX flavors (DPX.cpp:184-207) at sizes where words straddle slices and lines
    (synth.PIX_RGB12_PACKED_BE, 0), (synth.PIX_RGB12_PACKED_BE, synth.FLAG_VFLIP), (synth.PIX_RGBA10_FILLEDA_BE, 0), (synth.PIX_RGBA10_FILLEDA_LE, 0),
    (synth.PIX_RGBA12_PACKED_BE, 0), (synth.PIX_RGBA12_FILLEDA_BE, 0), (synth.PIX_RGBA12_FILLEDA_LE, 0), (synth.PIX_Y10_FILLEDA_BE, 0),
    (synth.PIX_Y10_FILLEDA_BE, synth.FLAG_ALTERN), (synth.PIX_Y10_FILLEDB_BE, 0), (synth.PIX_Y10_FILLEDB_BE, synth.FLAG_ALTERN),
    (synth.PIX_Y12_PACKED_BE, 0), (synth.PIX_Y12_PACKED_BE, synth.FLAG_VFLIP), (synth.PIX_EXR_RGB16, 0),
]


@pytest.mark.parametrize("pixfmt,flags", FLAVORS)
@pytest.mark.parametrize("w,h,slices", [(67, 29, 4), (130, 45, 9), (24, 10, 1)])
def test_bit_packed_flavors(built, pixfmt, flags, w, h, slices):
    """unpack kernel -> packet == oracle; device decoder + word-packing kernel -> the source bytes (padding bits zero)."""
    import torch
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    payloads = []
    for i, kind in enumerate(("film", "noise")):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, kind, seed=11 * pixfmt + i), pixfmt, pixfmt != synth.PIX_EXR_RGB16, flags)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1, flags)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=2, flags=flags)
    assert enc.config_record() == ob.config_record(p)
    packets = enc.encode_host(payloads)
    for f in range(2):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=2, flags=flags)
    dpk = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in packets]
    dout = [torch.full((len(x),), 0xAA, dtype=torch.uint8, device="cuda") for x in payloads]
    assert dec.decode_device([t.data_ptr() for t in dpk], [len(x) for x in packets], [t.data_ptr() for t in dout]) == 0
    for f in range(2):
        assert bytes(dout[f].cpu().numpy()) == payloads[f]
    enc.close(); dec.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RCGPU_SOAK_SEEDS", "12"))))      # soak: RCGPU_SOAK_SEEDS=300
def test_random_geometries_mixed_content(built, seed):
    """Stress for k_resolve's chunk pipeline: pictures made of flat patches (runs of zero residuals in one context), smooth ramps
    (many lanes per context -> rounds, forwarding between chunks) and noise, at random sizes / slice grids / segment counts, so that
    slices end in partial chunks of every length.  The packets are the oracle's byte for byte, decode to the pictures in the device decoder
    -- and in the REAL reference's (oracle/_ref/ref_ffv1_decode: ffv1_frame::Process itself, where its driver was built and the geometry is
    one its parser admits)."""
    import ext_streams
    import ref_decode
    m = ext_streams.mixed_content_stream(seed)
    w, h, pixfmt, nh, nv, segments, rc_span, payloads, line_bytes = m["w"], m["h"], m["pixfmt"], m["nh"], m["nv"], m["segments"], m["rc_span"], m["payloads"], m["line_bytes"]
    for ctx in (1, 2):
        p = ob.Params(w, h, pixfmt, nh, nv, 1, ctx)
        enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, ctx, max_batch=3, segments=segments, rc_span=rc_span)
        packets = enc.encode_host(payloads)
        for f in range(3):
            assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"context model {ctx}, frame {f}, {w}x{h} {nh}x{nv} segments {segments} rc_span {rc_span}"
        enc.close()
        if ref_decode.available() and m["reference_takes_it"]:                               # the REAL reference's decoder on the device's packets
            (frames,), lines = ref_decode.decode([(m["flavor"], 0, w, h, ob.config_record(p), packets)])
            assert ext_streams.reference_decodes_to(frames, m["tight"]), (lines, f"context model {ctx}, {w}x{h} {nh}x{nv}")
        dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, ctx, max_batch=3)        # and back through the device decoder
        assert dec.decode_host(packets, len(payloads[0])) == [bytes(x) for x in payloads], f"decoder, context model {ctx}, {w}x{h} {nh}x{nv}"
        dec.close()


@pytest.mark.parametrize("w,h,slices,pixfmt", [(2, 2, 1, synth.PIX_RGB16_BE), (3, 5, 1, synth.PIX_RGB10_FILLEDA_BE), (5, 4, 4, synth.PIX_RGB8), (64, 3, 1, synth.PIX_Y16_BE),
                                               (3, 64, 1, synth.PIX_RGBA16_BE), (9, 9, 4, synth.PIX_RGB12_PACKED_BE), (7, 3, 1, synth.PIX_Y10_FILLEDA_BE),
                                               (65, 2, 1, synth.PIX_EXR_RGB16), (6, 6, 4, synth.PIX_RGBA10_FILLEDA_BE)])
def test_tiny_pictures_and_extreme_values(built, w, h, slices, pixfmt):
    """Edge geometry: pictures smaller than one 64-symbol chunk, one-line slices, and the extreme sample values (all zero, all maximum,
    alternating) that fold residuals at the boundaries of the coded range."""
    import torch
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    maxv = (1 << bits) - 1
    comps = [np.zeros((h, w, nc), dtype=np.uint16), np.full((h, w, nc), maxv, dtype=np.uint16),
             ((np.add.outer(np.arange(h), np.arange(w)) & 1) * maxv).astype(np.uint16)[:, :, None].repeat(nc, axis=2),
             synth.components(w, h, nc, bits, "noise", seed=w * 7 + h)]
    payloads = []
    for c in comps:
        pl, line_bytes = synth.pack_payload(c, pixfmt, pixfmt != synth.PIX_EXR_RGB16)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=len(payloads))
    packets = enc.encode_host(payloads)
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=len(payloads))
    dpk = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in packets]
    dout = [torch.full((len(x),), 0x55, dtype=torch.uint8, device="cuda") for x in payloads]
    assert dec.decode_device([t.data_ptr() for t in dpk], [len(x) for x in packets], [t.data_ptr() for t in dout]) == 0
    for f in range(len(payloads)):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
        assert bytes(dout[f].cpu().numpy()) == payloads[f], f"frame {f}"
    enc.close(); dec.close()


@pytest.mark.parametrize("pixfmt,coder,ctx", [(synth.PIX_RGB16_BE, 1, 1), (synth.PIX_RGB10_FILLEDA_BE, 2, 1), (synth.PIX_Y16_BE, 1, 0), (synth.PIX_RGBA8, 2, 1)])
def test_level_1_single_slice_frames(built, pixfmt, coder, ctx):
    """-level 1 (what the reference asks for with -slices 1, Global.cpp:961-968): FFV1 version 1 -- the stream header inside every
    frame, one slice, no footer, no configuration record.  Same bytes as the oracle, whose version-1 frames the real reference decodes."""
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    w, h = 72, 40
    payloads = []
    for i in range(3):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, ["film", "noise", "flat"][i], seed=50 + i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, 1, 1, 0, ctx, 0, coder, 1)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 1, 1, 0, ctx, max_batch=3, coder=coder, level=1)
    assert enc.config_record() == b""
    packets = enc.encode_host(payloads)
    for f in range(3):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
    enc.close()
    # and back: the device decoder replays the in-band header, then the one slice
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, 1, 1, 0, ctx, max_batch=3, coder=coder, level=1)
    assert dec.decode_host(packets, len(payloads[0])) == [bytes(x) for x in payloads]
    broken = bytearray(packets[0]); broken[1] ^= 0x40                 # a different header is refused, not decoded as something else
    with pytest.raises(RuntimeError):
        dec.decode_host([bytes(broken)], len(payloads[0]))
    dec.close()
    with pytest.raises(RuntimeError):
        api.Ffv1Encoder(w, h, pixfmt, line_bytes, 2, 2, 0, ctx, level=1)
    with pytest.raises(RuntimeError):
        api.Ffv1Decoder(w, h, pixfmt, line_bytes, 2, 2, 0, ctx, level=1)
    with pytest.raises(RuntimeError, match="whole-slice"):          # the split coder is not offered for version 1 frames
        api.Ffv1Encoder(w, h, pixfmt, line_bytes, 1, 1, 0, ctx, level=1, rc_span=8)


def test_configurations_the_device_path_cannot_hold_are_refused(built):
    """Argument errors come back as errors (rcgpu_last_error), not as device faults: empty and over-large pictures, slice grids the
    reference's decoder would refuse, unknown layouts, a line stride shorter than a line."""
    ok = dict(width=64, height=48, pixfmt=synth.PIX_RGB16_BE, line_bytes=64 * 6, num_h=2, num_v=2)
    for bad in (dict(width=0), dict(height=0), dict(width=40000, height=40000, line_bytes=40000 * 6), dict(num_h=1, num_v=2), dict(num_h=65, num_v=1),
                dict(pixfmt=99), dict(line_bytes=64 * 6 - 2)):
        a = dict(ok, **bad)
        with pytest.raises(RuntimeError):
            api.Ffv1Encoder(a["width"], a["height"], a["pixfmt"], a["line_bytes"], a["num_h"], a["num_v"])
    for bad in (dict(width=0), dict(width=40000, height=40000, line_bytes=40000 * 6), dict(pixfmt=99), dict(line_bytes=64 * 6 - 2)):
        a = dict(ok, **bad)
        with pytest.raises(RuntimeError):
            api.Ffv1Decoder(a["width"], a["height"], a["pixfmt"], a["line_bytes"], a["num_h"], a["num_v"])


@pytest.mark.parametrize("slicecrc,context", [(0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("pixfmt,w,h,slices", [(synth.PIX_RGB16_BE, 200, 120, 6), (synth.PIX_RGB10_FILLEDA_BE, 130, 67, 9), (synth.PIX_RGBA16_LE, 64, 48, 4),
                                                (synth.PIX_Y16_BE, 80, 40, 4), (synth.PIX_RGB8, 257, 131, 16)])
def test_level3_without_slicecrc_and_with_the_3_input_model(built, pixfmt, w, h, slices, slicecrc, context):
    """`-slicecrc 0` (3-byte slice tail, ec = 0 in the record: k_footer's !ec branch, the decoder's tail = 3) and `-context 0`
    (3-input context model: is5 == 0 in k_model) at -level 3, both reachable from rawcooked's command line
    (Source/CLI/Global.cpp:337-485): multi-slice, several frames, segmented hand-over; bytes equal the oracle's and the device
    decoder rebuilds the payloads."""
    import torch
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    payloads = []
    for i in range(3):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, ["film", "noise", "flat"][i], seed=170 + i), pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, slicecrc, context)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, slicecrc, context, max_batch=3, segments=3)
    assert enc.config_record() == ob.config_record(p)
    packets = enc.encode_host(payloads)
    for f in range(3):
        assert packets[f] == ob.encode_payload(p, payloads[f], line_bytes), f"frame {f}"
        assert ob.decode_payload(p, packets[f], line_bytes) == payloads[f]
    cfg = api.config_from_record(enc.config_record(), w, h, pixfmt, line_bytes, context=context)
    assert (cfg.slicecrc, cfg.num_h_slices, cfg.num_v_slices) == (slicecrc, nh, nv)
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, slicecrc, context, max_batch=3)
    dpk = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in packets]
    dout = [torch.empty(len(x), dtype=torch.uint8, device="cuda") for x in payloads]
    assert dec.decode_device([t.data_ptr() for t in dpk], [len(x) for x in packets], [t.data_ptr() for t in dout]) == 0
    for f in range(3):
        assert bytes(dout[f].cpu().numpy()) == payloads[f]
    enc.close(); dec.close()


def test_a_second_bank_that_does_not_fit_leaves_nothing_behind(built):
    """rcgpu_ffv1_set_run_on allocates a second bank of per-batch buffers.  When the device has no room for it the call fails, everything it had
    allocated is given back (the memory is what the first batch's windows need), the encoder keeps coding one batch at a time, and a later
    attempt -- with room -- succeeds."""
    import torch
    w, h, pixfmt, nh, nv, n = 1024, 540, synth.PIX_RGB16_BE, 2, 2, 48
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    pls = [synth.pack_payload(synth.components(w, h, nc, bits, "film", seed=700 + i), pixfmt, True) for i in range(2)]
    line_bytes = pls[0][1]
    srcs = [pls[i % 2][0] for i in range(n)]
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    want = [ob.encode_payload(p, pls[i][0], line_bytes) for i in range(2)]
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n, rc_span=1)
    assert enc.encode_host(srcs[:4]) == [want[i % 2] for i in range(4)]                          # staging areas and windows exist now
    torch.cuda.synchronize()
    # the bank of 48 such frames is ~0.4 GB (0.7 before the slices' bytes moved into the symbol buffer).  The device is filled until an allocation
    # FAILS -- the free figure the runtime reports runs behind what it can still hand out when other tests have just freed memory -- and
    # one 64 MB block is given back
    fillers = [torch.empty(max(0, torch.cuda.mem_get_info()[0] - (1 << 30)), dtype=torch.uint8, device="cuda")]
    try:
        for _ in range(4096):
            fillers.append(torch.empty(64 << 20, dtype=torch.uint8, device="cuda"))
    except torch.OutOfMemoryError:
        pass
    fillers.pop()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    with pytest.raises(api.RcgpuError, match="second bank"):
        enc.set_run_on(True)
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0 - (64 << 20), (free0, free1)                                         # nothing of the half-made bank is held
    assert enc.encode_host(srcs[:6]) == [want[i % 2] for i in range(6)]                          # one batch at a time, as before
    del fillers
    torch.cuda.empty_cache()
    enc.set_run_on(True)                                                                        # with room: the mode is there
    assert enc.encode_host(srcs) == [want[i % 2] for i in range(n)]
    assert enc.error_flags() == 0
    enc.close()


OWN = 0x100          # RCGPU_FLAG_OWN_SLICE_BUFFERS


@pytest.mark.parametrize("w,h,pixfmt,slices,segments,rc_span",
                         [(600, 200, synth.PIX_RGB16_BE, 1, sg, sp) for sg in (1, 5, 32) for sp in (1, 8)] +
                         [(1024, 540, synth.PIX_RGB16_BE, 4, 32, 1), (1024, 540, synth.PIX_RGB16_BE, 4, 32, 8), (1024, 540, synth.PIX_RGB16_BE, 4, 5, 1),
                          (900, 400, synth.PIX_RGB10_FILLEDA_BE, 4, 5, 8), (900, 400, synth.PIX_RGB10_FILLEDA_BE, 4, 32, 1),
                          (1100, 600, synth.PIX_RGBA16_BE, 4, 32, 1), (1100, 600, synth.PIX_RGBA16_BE, 4, 1, 8)])
def test_slice_bytes_inside_the_symbol_buffer(built, w, h, pixfmt, slices, segments, rc_span):
    """Round 6: where every slice is large enough its coded bytes lie in its own area of the symbol buffer (the coder writes at most 3.4
    bytes where a 4-byte symbol lay that k_resolve has read), no slice byte buffers are allocated.  Same packets as with buffers of their
    own (RCGPU_FLAG_OWN_SLICE_BUFFERS) and as the oracle's, on noise -- the content that comes closest to the symbols -- and film, one
    batch at a time and in run-on mode; the estimate callers size batches with says what the overlay saves."""
    import ctypes as C
    import torch
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    nh, nv = api.slices_to_grid(slices)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    pls = [synth.pack_payload(synth.components(w, h, nc, bits, ["noise", "film", "noise"][i], seed=300 + i), pixfmt, True) for i in range(3)]
    line_bytes = pls[0][1]
    pls = [x[0] for x in pls]
    want = [ob.encode_payload(p, pl, line_bytes) for pl in pls]
    L = api.lib()
    L.rcgpu_ffv1_device_bytes_per_frame.restype, L.rcgpu_ffv1_device_bytes_per_frame.argtypes = C.c_uint64, [C.c_void_p, C.c_int]
    per = {}
    for flags in (0, OWN):
        enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3, segments=segments, rc_span=rc_span, flags=flags)
        per[flags] = [L.rcgpu_ffv1_device_bytes_per_frame(C.byref(enc.cfg), r) for r in (0, 1)]
        assert enc.encode_host(pls) == want, f"flags {flags:#x}"
        assert enc.error_flags() == 0
        enc.set_run_on(True)
        for _ in range(2):
            assert enc.encode_host(pls[::-1]) == want[::-1], f"flags {flags:#x}, run-on"
        assert enc.error_flags() == 0
        enc.close()
    raw = len(pls[0])
    assert per[OWN][0] - per[0][0] >= raw * 3 // 2 and per[OWN][1] - per[0][1] >= 2 * (raw * 3 // 2)      # one set of slice buffers, two in run-on mode


def test_bytes_that_reach_unread_symbols_are_reported(built):
    """The overlay's own limit: while segment j is coded the bytes must stay below the symbols of segment j + 1, four bytes per sample from the
    slice's start.  Real content cannot get there (16-bit noise with untrained states codes to 3.4 bytes per sample), so the limit is halved
    (slice_buffer_div = 2): a picture whose first lines are noise and whose rest is flat fits its buffer many times over in the end, yet its
    first segment runs past the halved limit -- reported (the error word, never silently wrong bytes); with buffers of their own the same
    configuration codes it, and so does the overlay with the true limit."""
    import torch
    w, h, pixfmt = 600, 200, synth.PIX_RGB16_BE
    comp = synth.components(w, h, 3, 16, "flat", seed=3).copy()
    comp[:6] = synth.components(w, h, 3, 16, "noise", seed=4)[:6]            # the first of 32 segments (200 / 32 lines) and no more
    pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
    p = ob.Params(w, h, pixfmt, 1, 1, 1, 1)
    want = ob.encode_payload(p, pl, line_bytes)
    assert len(want) < len(pl) // 8
    d = torch.frombuffer(bytearray(pl), dtype=torch.uint8).cuda()
    for flags, div, ok in ((0, 0, True), (OWN, 2, True), (0, 2, False)):
        enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 1, 1, 1, 1, max_batch=1, segments=32, flags=flags, slice_buffer_div=div)
        pk = torch.zeros(enc.max_packet, dtype=torch.uint8, device="cuda")
        sz = torch.zeros(1, dtype=torch.int64, device="cuda")
        enc.encode_device([d.data_ptr()], pk.data_ptr(), enc.max_packet, sz.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if ok:
            assert enc.error_flags() == 0 and bytes(pk[:int(sz[0])].cpu().numpy()) == want, (flags, div)
        else:
            with pytest.raises(api.RcgpuError, match="outgrew"):
                enc.error_flags()
        enc.close()


def test_batch_intervals_time_the_steps_of_a_run_on_caller(built):
    """rcgpu_ffv1_batch_intervals: the device time between the ends of consecutive batches, from events the library records behind each batch's
    last kernel -- what bench.py's `step_ms` is made of.  Five batches in run-on mode: four intervals, oldest first, all positive, and together no
    longer than the wall clock around the five calls and the join."""
    import time
    import torch
    w, h, pixfmt = 512, 270, synth.PIX_RGB16_BE
    pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=9), pixfmt, True)
    n = 8
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 1, 1, 1, 1, max_batch=n)
    assert enc.batch_intervals() == []
    d = [torch.frombuffer(bytearray(pl), dtype=torch.uint8).cuda() for _ in range(n)]
    stride = (enc.max_packet + 255) & ~255
    pk = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
    sz = torch.zeros(n, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    enc.set_run_on(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        enc.encode_device([t.data_ptr() for t in d], pk.data_ptr(), stride, sz.data_ptr(), st)
    enc.join(st); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    iv = enc.batch_intervals()
    assert len(iv) == 4 and all(x > 0 for x in iv) and sum(iv) <= wall, (iv, wall)
    assert enc.batch_intervals(2) == iv[-2:]
    assert enc.error_flags() == 0
    enc.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RCGPU_SOAK_OVERLAY", "6"))))      # soak: RCGPU_SOAK_OVERLAY=60
def test_overlay_random_geometries(built, seed):
    """The overlay at random: pictures whose slices all hold 64 K samples or more (so that their coded bytes lie inside the symbol buffer), random
    size, pixel layout, slice grid, segment count and coder mapping, every frame a patchwork of noise, film and flat areas -- noise in a slice's
    first lines is what brings the bytes closest to the symbols not read yet.  One batch at a time and in run-on mode: the oracle's packets."""
    rng = np.random.default_rng(9000 + seed)
    pixfmt = [synth.PIX_RGB16_BE, synth.PIX_RGB16_LE, synth.PIX_RGB10_FILLEDA_BE, synth.PIX_RGBA16_BE, synth.PIX_Y16_BE, synth.PIX_RGB8, synth.PIX_RGB12_PACKED_BE][int(rng.integers(0, 7))]
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    slices = [1, 4, 6, 9][int(rng.integers(0, 4))]
    nh, nv = api.slices_to_grid(slices)
    need = 66000 // nc + 1                                   # pixels per slice
    for _ in range(100):
        w, h = int(rng.integers(260, 1500)), int(rng.integers(160, 900))
        if (w // nh) * (h // nv) >= need and w * h * nc <= 2_400_000:
            break
    else:
        pytest.skip("no geometry drawn")
    segments = int([1, 3, 7, 32, 0][int(rng.integers(0, 5))])
    rc_span = int([1, 8, 64][int(rng.integers(0, 3))])
    payloads = []
    for f in range(3):
        comp = synth.components(w, h, nc, bits, "film", seed=seed * 10 + f).copy()
        for _ in range(int(rng.integers(1, 6))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = min(w, x0 + int(rng.integers(8, w))), min(h, y0 + int(rng.integers(2, h)))
            kind = ["noise", "flat", "noise"][int(rng.integers(0, 3))]
            comp[y0:y1, x0:x1] = synth.components(w, h, nc, bits, kind, seed=seed * 100 + f)[y0:y1, x0:x1]
        if f == 0:
            comp[:max(1, h // (8 * nv))] = synth.components(w, h, nc, bits, "noise", seed=seed + 77)[:max(1, h // (8 * nv))]      # noise at the top of the first slice row
        pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
        payloads.append(pl)
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    what = f"{w}x{h} pixfmt {pixfmt} {nh}x{nv} segments {segments} rc_span {rc_span}"
    L = api.lib()
    import ctypes as C
    probe = lambda fl: L.rcgpu_ffv1_device_bytes_per_frame(C.byref(api.Ffv1Config(w, h, pixfmt, line_bytes, nh, nv, 1, 1, 3, 0, segments, fl, 1, 3, rc_span, 0)), 0)
    assert probe(0) < probe(OWN), "the overlay is off for " + what
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3, segments=segments, rc_span=rc_span)
    assert enc.encode_host(payloads) == want, what
    enc.set_run_on(True)
    for k in range(3):
        order = [(k + i) % 3 for i in range(3)]
        assert enc.encode_host([payloads[i] for i in order]) == [want[i] for i in order], what + " run-on"
    assert enc.error_flags() == 0
    enc.close()
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3)
    assert dec.decode_host(want, len(payloads[0])) == [bytes(x) for x in payloads], what
    dec.close()
    # ... and with the compact context model (states in LDS, k_resolve<true>): the same overlay under the other kernel
    p2 = ob.Params(w, h, pixfmt, nh, nv, 1, 2)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 2, max_batch=3, segments=segments, rc_span=rc_span)
    assert enc.encode_host(payloads[:2]) == [ob.encode_payload(p2, pl, line_bytes) for pl in payloads[:2]], what + " compact model"
    assert enc.error_flags() == 0
    enc.close()
