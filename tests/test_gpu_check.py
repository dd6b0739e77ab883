"""GPU: the --check half (BASELINE config 5) -- device FFV1 decode + inverse transform, byte compare and MD5.
Bit-exact against the source payloads, against the committed golden packets and against hashlib."""
import hashlib
import json
import os

import numpy as np

import pytest
import torch

import oracle_binding as ob
from rawcooked_amd import api, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))


def dev(b: bytes):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("v", VEC["ffv1"], ids=lambda v: v["name"])
def test_decode_golden_packets(built, v):
    """Packets the real reference accepted -> device decoder -> the original payload bytes."""
    n = len(v["frames"])
    payloads = [open(os.path.join(G, f["payload"]), "rb").read() for f in v["frames"]]
    packets = [open(os.path.join(G, f["packet"]), "rb").read() for f in v["frames"]]
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["num_h"], v["num_v"], 1, 1, max_batch=n, flags=v["flags"], coder=v["coder"])
    dpk = [dev(p) for p in packets]
    dout = [torch.full((len(p),), 0xAA, dtype=torch.uint8, device="cuda") for p in payloads]
    flags = dec.decode_device([t.data_ptr() for t in dpk], [len(p) for p in packets], [t.data_ptr() for t in dout])
    assert flags == 0
    for i in range(n):
        assert bytes(dout[i].cpu().numpy()) == payloads[i], f"frame {i}"
    dec.close()


def _ext_record(v):
    return open(os.path.join(G, v["config_record_file"]), "rb").read() if "config_record_file" in v else bytes.fromhex(v.get("config_record", ""))


@pytest.mark.parametrize("v", [v for v in VEC["ffv1_ext"] if "unsup" not in v["name"]], ids=lambda v: v["name"])
def test_decode_streams_other_encoders_could_write(built, v):
    """What parameters::Parse accepts and FFmpeg's defaults never produce (FFV1_Parameters.cpp:23-183,206-253; FFV1_Slice.cpp:158-168): 1 to 8
    table sets of arbitrary tables, a set of its own per plane group, a transmitted transition table nobody ships, coded initial states, the
    version 0 / 1 header inside the frame.  Each stream was checked and decoded by the real reference (tests/golden/make_golden.py); the device
    decoder knows the picture's geometry and reads everything else from CodecPrivate and the first packet -- to the same payload bytes, with
    the full window and with samples forced down the careful path."""
    n = len(v["frames"])
    payloads = [open(os.path.join(G, f["payload"]), "rb").read() for f in v["frames"]]
    packets = [open(os.path.join(G, f["packet"]), "rb").read() for f in v["frames"]]
    stream = api.Ffv1Stream(_ext_record(v), packets[0])
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], max_batch=n + 1, stream=stream)
    stream.close()                                              # the decoder keeps what it needs
    dpk = [dev(p) for p in packets]
    for cap in (7, 2):
        dec.debug_window(cap)
        dout = [torch.full((len(p),), 0xAA, dtype=torch.uint8, device="cuda") for p in payloads]
        assert dec.decode_device([t.data_ptr() for t in dpk], [len(p) for p in packets], [t.data_ptr() for t in dout]) == 0
        for i in range(n):
            assert bytes(dout[i].cpu().numpy()) == payloads[i], f"window {cap}: frame {i}"
    assert dec.decode_host(packets[::-1], len(payloads[0])) == payloads[::-1]            # a second batch: the states are set anew for every batch
    bad = bytearray(packets[0])
    bad[len(bad) // 3] ^= 0x08
    if v["ec"]:
        with pytest.raises(api.RcgpuError, match="undecodable"):
            dec.decode_host([bytes(bad)], len(payloads[0]))
    else:                                                       # no slice CRC: the decoder may run to the end, but not to the same picture
        try:
            assert dec.decode_host([bytes(bad)], len(payloads[0])) != [payloads[0]]
        except api.RcgpuError:
            pass
    dec.close()


def test_streams_the_device_does_not_take_are_told_apart_from_broken_ones(built):
    """Two valid streams (the real reference decodes both, tests/golden/make_golden.py) outside the device decoder: intra = 0 is refused when
    the decoder is made (RCGPU_FFV1_UNSUPPORTED); slices that name other table sets than the first slice did -- quant_table_set_index is a
    per-slice field, FFV1_Slice.cpp:158-168 -- are found while decoding and reported like any undecodable frame (header flag 32).  Either
    way a binding hands the frames to its own decoder (route C: test_gpu_e2e.py)."""
    by = {v["name"]: v for v in VEC["ffv1_ext"]}
    v = by["ext_unsup_intra0_64x48"]
    pk = open(os.path.join(G, v["frames"][0]["packet"]), "rb").read()
    s = api.Ffv1Stream(_ext_record(v), pk)
    with pytest.raises(api.RcgpuUnsupported):
        api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], max_batch=1, stream=s)
    v = by["ext_unsup_per_slice_sets_64x48"]
    pk = open(os.path.join(G, v["frames"][0]["packet"]), "rb").read()
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], max_batch=1, stream=api.Ffv1Stream(_ext_record(v), pk))
    with pytest.raises(api.RcgpuError, match="undecodable.*0x20"):
        dec.decode_host([pk], len(open(os.path.join(G, v["frames"][0]["payload"]), "rb").read()))
    dec.close()


@pytest.mark.parametrize("window", [1, 2, 3, 5])
@pytest.mark.parametrize("v", [v for v in VEC["ffv1"] if v["name"] in ("dpx_rgb16be_64x48", "dpx_rgb10be_50x38", "dpx_rgba12packed_50x38", "dpx_y16be_40x24", "tiff_rgb8_40x30",
                                                                       "exr_rgb16_72x40", "dpx_rgb16be_coder2_72x40")], ids=lambda v: v["name"])
def test_samples_that_outrun_their_window_are_decoded_again_carefully(built, v, window):
    """The sample decoder takes bytes out of a 7-byte window without looking; a sample that took more than the window held (the sentinel
    was consumed: an all-zero window) is decoded again by the decoder that looks before it takes.  Real pictures almost never get there,
    so the tests shrink the window (rcgpu_ffv1_decoder_debug_window): with one byte nearly every sample of noisy content goes the careful
    way -- and the payloads must be the same."""
    n = len(v["frames"])
    payloads = [open(os.path.join(G, f["payload"]), "rb").read() for f in v["frames"]]
    packets = [open(os.path.join(G, f["packet"]), "rb").read() for f in v["frames"]]
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["num_h"], v["num_v"], 1, 1, max_batch=n, flags=v["flags"], coder=v["coder"])
    dpk = [dev(p) for p in packets]
    careful = {}
    for cap in (7, window):
        dec.debug_window(cap)
        dout = [torch.full((len(p),), 0xAA, dtype=torch.uint8, device="cuda") for p in payloads]
        assert dec.decode_device([t.data_ptr() for t in dpk], [len(p) for p in packets], [t.data_ptr() for t in dout]) == 0
        for i in range(n):
            assert bytes(dout[i].cpu().numpy()) == payloads[i], f"window {cap}: frame {i}"
        careful[cap] = dec.debug_careful()
    assert careful[7] <= 2 and careful[window] >= careful[7], careful       # with the full window: (almost) never in these small pictures
    if window <= 2:
        assert careful[window] > 0, careful           # ... and a window of one or two bytes cannot hold a 10..16-bit sample's decisions
    dec.close()


def test_a_worst_case_sample_outruns_the_full_window(built):
    """One slice of uniform 16-bit noise after a long flat run: the contexts' states have adapted to "no change", and the first noisy
    samples cost far more than the seven bytes of a window.  With the FULL window, the careful path runs, and the picture is right."""
    import numpy as np
    w, h, pixfmt = 256, 64, synth.PIX_RGB16_BE
    rng = np.random.default_rng(5)
    comp = np.full((h, w, 3), 1000, dtype=np.uint16)
    comp[8:, :, :] = rng.integers(0, 65536, size=(h - 8, w, 3), dtype=np.uint16)        # flat rows train the states, then noise
    comp[::2, ::7, :] = 65535; comp[1::2, 3::5, :] = 0                                      # and spikes of full amplitude
    pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
    p = ob.Params(w, h, pixfmt, 1, 1, 1, 1)
    pk = ob.encode_payload(p, pl, line_bytes)
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, 1, 1, 1, 1, max_batch=1)
    assert dec.decode_host([pk], len(pl))[0] == pl
    n7 = dec.debug_careful()
    dec.debug_window(4)
    assert dec.decode_host([pk], len(pl))[0] == pl and dec.debug_careful() >= n7
    dec.close()


def test_encode_then_decode_on_device_and_verify(built):
    """encode -> decode round trip entirely in HBM at a non-trivial size, verified with the device compare and MD5."""
    w, h, pixfmt, nh, nv, n = 640, 360, synth.PIX_RGB16_BE, 4, 4, 5
    srcs = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=9 + i), pixfmt, True)
        srcs.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    dsrc = [dev(p) for p in srcs]
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
        assert api.compare_device(dout[i].data_ptr(), dsrc[i].data_ptr(), len(srcs[i])) == -1          # FileWriter.cpp:448-463
    md5s = api.md5_device([t.data_ptr() for t in dout], [len(s) for s in srcs])
    assert md5s == [hashlib.md5(s).digest() for s in srcs]                                            # FileWriter.cpp:596-727
    # negative controls: a flipped payload byte is located, a flipped packet byte is refused
    assert api.compare_device_batch([t.data_ptr() for t in dout], [t.data_ptr() for t in dsrc], [len(s) for s in srcs]) == [-1] * n
    dout[2][12345] ^= 1
    assert api.compare_device(dout[2].data_ptr(), dsrc[2].data_ptr(), len(srcs[2])) == 12345
    dout[0][len(srcs[0]) - 1] ^= 0x80                      # the last byte, and views that start at odd addresses (byte-wise path)
    want = [len(srcs[0]) - 1, -1, 12345] + [-1] * (n - 3)
    assert api.compare_device_batch([t.data_ptr() for t in dout], [t.data_ptr() for t in dsrc], [len(s) for s in srcs]) == want[:n]
    assert api.compare_device_batch([dout[2].data_ptr() + 1], [dsrc[2].data_ptr() + 1], [len(srcs[2]) - 1]) == [12344]
    dout[0][len(srcs[0]) - 1] ^= 0x80
    dpk[stride + sizes[1] // 2] ^= 0x10
    with pytest.raises(api.RcgpuError, match="undecodable"):
        dec.decode_device([dpk.data_ptr() + i * stride for i in range(n)], sizes, [t.data_ptr() for t in dout])
    enc.close(); dec.close()


def test_md5_device_edges(built):
    msgs = [b"", b"a", b"abc", b"x" * 55, b"y" * 56, b"z" * 63, b"q" * 64, bytes(range(256)) * 5, os.urandom(100003)]
    bufs = [dev(m if m else b"\0") for m in msgs]
    got = api.md5_device([t.data_ptr() for t in bufs], [len(m) for m in msgs])
    assert got == [hashlib.md5(m).digest() for m in msgs]


def test_many_small_verification_calls_on_a_side_stream(built):
    """A --check binding calls the compare and the MD5 once per small batch, hundreds of times, on its own stream."""
    st = torch.cuda.Stream()
    a = dev(os.urandom(40000)); b = a.clone()
    b[777] ^= 1
    ha, hb = bytes(a.cpu().numpy()), bytes(b.cpu().numpy())
    want = hashlib.md5(ha).digest()
    shifted = next(i for i in range(39000) if ha[i] != hb[1000 + i])
    for k in range(300):
        assert api.compare_device_batch([a.data_ptr()], [b.data_ptr()], [40000], st.cuda_stream) == [777]
        assert api.compare_device_batch([a.data_ptr(), a.data_ptr()], [a.data_ptr(), b.data_ptr() + 1000], [40000, 39000], st.cuda_stream) == [-1, shifted]
        if k % 10 == 0:
            assert api.md5_device([a.data_ptr()], [40000], st.cuda_stream) == [want]


def test_md5_device_at_any_address(built):
    """Views that start at every offset modulo 16 (whole 16-byte reads, word reads, funnel-shifted words) and lengths around the
    block, the padding and the extra word the shifted path reads."""
    blob = os.urandom(70000)
    t = dev(blob)
    views = [(o, n) for o in range(17) for n in (0, 1, 63, 64, 67, 68, 69, 127, 128, 131, 132, 4096, 4099, 65000)]
    got = api.md5_device([t.data_ptr() + o for o, _ in views], [n for _, n in views])
    assert got == [hashlib.md5(blob[o:o + n]).digest() for o, n in views]


def test_check_with_the_payloads_kept_on_the_device(built):
    """decode_keep / kept_to_host / verify_kept: what frame_writer does with every rebuilt file -- MD5 of Pre + plane + Post
    (FileWriter.cpp:709-724) and the comparison with the file on disk (:581-589, length :200) -- for a batch, without the payloads
    coming back."""
    w, h, pixfmt, nh, nv, n = 320, 180, synth.PIX_RGB16_BE, 4, 4, 7
    srcs = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=40 + i), pixfmt, True)
        srcs.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    packets = enc.encode_host(srcs)
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    dec.decode_keep(packets)
    assert [dec.kept_to_host(i, len(srcs[i])) for i in range(n)] == srcs
    before = [os.urandom(k) for k in (2048, 0, 7, 8192, 65536, 1, 2050)]
    after = [os.urandom(k) for k in (0, 33, 0, 65536, 5, 0, 100)]
    files = [before[i] + srcs[i] + after[i] for i in range(n)]
    disk = list(files)
    disk[1] = files[1][:-1] + bytes([files[1][-1] ^ 1])                 # last byte of the bytes after the payload
    disk[2] = bytes([files[2][0] ^ 0x40]) + files[2][1:]                # first byte of the bytes before it
    disk[3] = files[3][:8192 + 4321] + bytes([files[3][8192 + 4321] ^ 2]) + files[3][8192 + 4322:]   # inside the payload
    disk[4] = files[4][:-3]                                             # the file on disk is shorter
    disk[5] = files[5] + b"\0"                                          # ... longer
    disk[6] = None                                                      # nothing to compare with: hash only
    order = [3, 0, 6, 1, 5, 2, 4]
    got = dec.verify_kept([{"slot": i, "before": before[i], "after": after[i], "on_disk": disk[i]} for i in order])
    want_diff = {0: -1, 1: len(files[1]) - 1, 2: 0, 3: 8192 + 4321, 4: len(files[4]) - 3, 5: len(files[5]), 6: -1}
    for k, i in enumerate(order):
        assert got[k] == (hashlib.md5(files[i]).digest(), want_diff[i]), i
    # compare only; a second batch replaces the slots; bad arguments are refused
    got = dec.verify_kept([{"slot": 0, "before": b"", "after": b"", "on_disk": srcs[0], "md5": False}])
    assert got == [(bytes(16), -1)]
    dec.decode_keep(packets[4:6])
    assert dec.verify_kept([{"slot": 1, "before": b"ab", "after": b"c", "on_disk": b"ab" + srcs[5] + b"c"}]) == [(hashlib.md5(b"ab" + srcs[5] + b"c").digest(), -1)]
    # the packets by their place in a file, the files on disk by their names (read with pread() instead of through mappings)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        blob, offs = b"head", []
        for pk in packets:
            offs.append(len(blob)); blob += pk + b"--"
        open(os.path.join(tmp, "all.bin"), "wb").write(blob)
        fd = os.open(os.path.join(tmp, "all.bin"), os.O_RDONLY)
        dec.decode_keep_fd(fd, offs, [len(pk) for pk in packets])
        assert [dec.kept_to_host(i, len(srcs[i])) for i in range(n)] == srcs
        with pytest.raises(api.RcgpuError, match="file ends"):
            dec.decode_keep_fd(fd, [offs[0], len(blob) - 10], [len(packets[0]), len(packets[1])])
        os.close(fd)
        dec.decode_keep(packets)
        named = []
        for i in range(n):
            if disk[i] is not None:
                open(os.path.join(tmp, "f%d" % i), "wb").write(disk[i])
                named.append(i)
        got = dec.verify_kept([{"slot": i, "before": before[i], "after": after[i], "on_disk_path": os.path.join(tmp, "f%d" % i)} for i in named])
        assert got == [(hashlib.md5(files[i]).digest(), want_diff[i]) for i in named]
        with pytest.raises(api.RcgpuError, match="cannot open"):
            dec.verify_kept([{"slot": 0, "before": b"", "after": b"", "on_disk_path": os.path.join(tmp, "nothing")}])
    # in two calls: the next batch is decoded (into a second set of slots) while the files of this one are hashed
    dec.decode_keep(packets[:4])
    dec.verify_kept([{"slot": i, "before": before[i], "after": after[i], "on_disk": disk[i]} for i in (2, 0, 3)], begin_only=True)
    with pytest.raises(api.RcgpuError, match="waiting"):
        dec.verify_kept([{"slot": 1, "before": b"", "after": b""}])
    dec.decode_keep(packets[5:7])
    assert dec.kept_to_host(1, len(srcs[6])) == srcs[6]
    dec.decode_keep(packets[5:7])                                       # ... and once more: the waiting set is left alone
    assert dec.verify_kept_end() == [(hashlib.md5(files[i]).digest(), want_diff[i]) for i in (2, 0, 3)]
    with pytest.raises(api.RcgpuError, match="no verification"):
        dec.verify_kept_end()
    assert dec.verify_kept([{"slot": 0, "before": b"x", "after": b"", "on_disk": b"x" + srcs[5]}]) == [(hashlib.md5(b"x" + srcs[5]).digest(), -1)]
    dec.decode_keep(packets[4:6])
    for bad in ([{"slot": 2, "before": b"", "after": b""}], [{"slot": 0, "before": bytes(65537), "after": b""}],
                [{"slot": 0, "before": b"", "after": b""}, {"slot": 0, "before": b"", "after": b""}]):
        with pytest.raises(api.RcgpuError):
            dec.verify_kept(bad)
    with pytest.raises(api.RcgpuError, match="undecodable"):
        dec.decode_keep([packets[0], packets[1][:100] + bytes([packets[1][100] ^ 8]) + packets[1][101:]])
    with pytest.raises(api.RcgpuError):
        dec.kept_to_host(0, len(srcs[0]))                               # nothing is kept after a failed batch
    enc.close(); dec.close()


def test_a_batch_decoded_ahead(built):
    """rcgpu_ffv1_decoder_decode_keep_hint: the next batch is decoded by a thread of the library's own while the caller has the current one
    hashed and compared; the decode_keep that asks for exactly that batch adopts its slots, any other batch (or a failed hint) is decoded
    as if nothing had been hinted, and three sets of slots go round: under verification, current, ahead."""
    w, h, pixfmt, nh, nv, n = 320, 180, synth.PIX_RGB16_BE, 4, 4, 12
    srcs = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=90 + i), pixfmt, True)
        srcs.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    packets = enc.encode_host(srcs)
    enc.close()
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=4)
    B = [api.Ffv1Decoder.PacketBatch(packets[k:k + 4]) for k in (0, 4, 8)]
    files = lambda k: [{"slot": i, "before": b"pre", "after": b"post", "on_disk": b"pre" + srcs[k + i] + b"post"} for i in range(4)]
    want = lambda k: [(hashlib.md5(b"pre" + srcs[k + i] + b"post").digest(), -1) for i in range(4)]
    # the steady state of route C: batch k current, k+1 ahead, k-1 under verification
    dec.decode_keep(B[0]); dec.decode_keep_hint(B[1])
    assert [dec.kept_to_host(i, len(srcs[i])) for i in range(4)] == srcs[0:4]           # the current batch's slots, with the next one in flight
    dec.verify_kept(files(0), begin_only=True)
    dec.decode_keep(B[1])                                                                # adopted
    dec.decode_keep_hint(B[2])                                                           # third set: batch 0 is still being hashed
    assert dec.verify_kept_end() == want(0)
    assert [dec.kept_to_host(i, len(srcs[4 + i])) for i in range(4)] == srcs[4:8]
    dec.verify_kept(files(4), begin_only=True)
    dec.decode_keep(B[2])
    assert dec.verify_kept_end() == want(4)
    assert dec.verify_kept(files(8)) == want(8)
    # a hint for another batch than the one asked for next is dropped; so is one for a batch with a broken packet (and the error is the asker's)
    dec.decode_keep_hint(B[0])
    dec.decode_keep(B[1])
    assert [dec.kept_to_host(i, len(srcs[4 + i])) for i in range(4)] == srcs[4:8]
    bad = api.Ffv1Decoder.PacketBatch([packets[0], packets[1][:100] + bytes([packets[1][100] ^ 8]) + packets[1][101:]])
    dec.decode_keep_hint(bad)
    with pytest.raises(api.RcgpuError, match="undecodable"):
        dec.decode_keep(bad)
    dec.decode_keep_hint(B[2])
    assert dec.decode_host(packets[0:2], len(srcs[0])) == srcs[0:2]                                    # decode_host drops the batch in flight
    dec.decode_keep(B[2])
    assert [dec.kept_to_host(i, len(srcs[8 + i])) for i in range(4)] == srcs[8:12]
    dec.decode_keep_hint(B[0])                                                           # destroyed with a batch in flight
    dec.close()


@pytest.mark.parametrize("name", ["dpx_rgb16be_64x48", "dpx_rgb10be_50x38", "dpx_rgba12packed_50x38", "exr_rgb16_72x40"])
def test_corrupted_packets_never_hang_or_fault(built, name):
    """Robustness of the device decoder: random byte flips, truncations and garbage tails in reference-blessed packets.  With slice
    CRCs the damage must be reported; in every case the call returns (no hang, no fault) and later calls still decode correctly."""
    import numpy as np
    v = [x for x in VEC["ffv1"] if x["name"] == name][0]
    good = open(os.path.join(G, v["frames"][0]["packet"]), "rb").read()
    payload = open(os.path.join(G, v["frames"][0]["payload"]), "rb").read()
    rng = np.random.default_rng(5)
    variants = []
    for k in range(24 * int(os.environ.get("RCGPU_SOAK_CORRUPT", "1"))):      # soak: RCGPU_SOAK_CORRUPT=40
        b = bytearray(good)
        kind = k % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
        elif kind == 1:
            b = b[:int(rng.integers(1, len(b)))]
        elif kind == 2:
            at = int(rng.integers(0, len(b)))
            b[at:] = bytes(rng.integers(0, 256, size=len(b) - at, dtype=np.uint8))
        else:
            b[-8:] = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))      # slice size / CRC of the last slice
        variants.append(bytes(b))
    n = len(variants)
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["num_h"], v["num_v"], 1, 1, max_batch=n, flags=v["flags"])
    dpk = [dev(p) for p in variants]
    dout = [torch.zeros(len(payload), dtype=torch.uint8, device="cuda") for _ in range(n)]
    with pytest.raises(RuntimeError):
        dec.decode_device([t.data_ptr() for t in dpk], [len(p) for p in variants], [t.data_ptr() for t in dout])
    torch.cuda.synchronize()
    # one at a time: each damaged packet is refused on its own, and an undamaged one still decodes afterwards
    for i in range(n):
        try:
            flags = dec.decode_device([dpk[i].data_ptr()], [len(variants[i])], [dout[i].data_ptr()])
        except RuntimeError:
            flags = 1
        assert flags != 0 or bytes(dout[i].cpu().numpy()) == payload          # a flip may land in padding the decoder never reads
    g = dev(good)
    assert dec.decode_device([g.data_ptr()], [len(good)], [dout[0].data_ptr()]) == 0
    assert bytes(dout[0].cpu().numpy()) == payload
    dec.close()


@pytest.mark.parametrize("pixfmt,ctx", [(synth.PIX_RGB16_BE, 1), (synth.PIX_RGB10_FILLEDA_BE, 1), (synth.PIX_RGBA16_LE, 2), (synth.PIX_Y16_BE, 1), (synth.PIX_RGB8, 1)])
def test_damaged_streams_without_slice_crc_decode_like_the_reference(built, pixfmt, ctx):
    """Without slice CRCs (ec = 0) most damage is invisible to a decoder: it decodes SOMETHING.  Whatever the device decodes without a
    complaint must be what the REAL reference's decoder makes of the same bytes (oracle/_ref/ref_ffv1_decode: ffv1_frame::Process) -- the
    byte compare with the source file that follows in --check then says the same in both.  Flipped bytes, truncations and garbage tails;
    RCGPU_SOAK_CORRUPT scales the number of variants."""
    import ext_streams
    import ref_decode
    if not ref_decode.available():
        pytest.skip("oracle/_ref/ref_ffv1_decode not built (needs /root/reference)")
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    w, h, nh, nv = 64, 48, 2, 2
    comp = synth.components(w, h, nc, bits, "film", seed=77)
    payload, line_bytes = synth.pack_payload(comp, pixfmt, True)
    tight = synth.pack_payload(comp, pixfmt, False)[0]
    p = ob.Params(w, h, pixfmt, nh, nv, 0, ctx)
    good, rec = ob.encode_payload(p, payload, line_bytes), ob.config_record(p)
    rng = np.random.default_rng(6 + pixfmt)
    variants = [good]
    for k in range(48 * int(os.environ.get("RCGPU_SOAK_CORRUPT", "1"))):
        b = bytearray(good)
        kind = k % 4
        if kind in (0, 1):
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b) - 3))] ^= int(rng.integers(1, 256))          # (the last three bytes: the last slice's size)
        elif kind == 2:
            at = int(rng.integers(len(b) // 2, len(b) - 3))
            b[at:-3] = bytes(rng.integers(0, 256, size=len(b) - 3 - at, dtype=np.uint8))
        else:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(b))
    res, lines = ref_decode.decode([(ext_streams.flavor_of(pixfmt), 0, w, h, rec, [pk]) for pk in variants])       # (a decoder each: its first complaint stays with it)
    frames = [r[0] for r in res]
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 0, ctx, max_batch=1)
    quiet = same = 0
    for i, pk in enumerate(variants):
        try:
            got = dec.decode_host([pk], len(payload))[0]
        except RuntimeError:
            continue                                             # a complaint: the frame is the reference's own decoder's (route C)
        quiet += 1
        verdict, ref_bytes = frames[i]
        # the device's lines carry the DPX room at their end, the reference's plane does not (ext_streams.reference_decodes_to): line by line
        tl = len(tight) // h
        got_tight = b"".join(got[y * line_bytes:y * line_bytes + tl] for y in range(h))
        assert verdict == 0 and ref_bytes[:len(tight)] == got_tight, f"variant {i}: the device decoded it without a complaint, the reference {'complains' if verdict else 'decodes other bytes'}"
        same += 1
    # (the reference's end-of-slice tests -- bytes left over, bytes missing: FFV1-SLICE-JUNK, FFV1-SLICE-SliceContent -- catch nearly all damage even
    # without a CRC, and so do the device's: the quiet set is small, the undamaged packet always in it)
    assert quiet >= 1 and same == quiet and frames[0][0] == 0 and frames[0][1][:len(tight)] == tight
    dec.close()


@pytest.mark.parametrize("name", ["dpx_rgb16be_64x48", "dpx_rgb12packed_56x38", "dpx_rgb10be_coder2_50x38", "tiff_rgba16le_40x30"])
def test_decode_host_from_codec_private(built, name):
    """The caller of the --check half holds Matroska blocks and a CodecPrivate in host memory (ffv1_wrapper::Process/OutOfBand,
    Wrapper.cpp:115-128): configuration from the record, packets up, payloads back."""
    v = [x for x in VEC["ffv1"] if x["name"] == name][0]
    payloads = [open(os.path.join(G, f["payload"]), "rb").read() for f in v["frames"]]
    packets = [open(os.path.join(G, f["packet"]), "rb").read() for f in v["frames"]]
    cfg = api.config_from_record(bytes.fromhex(v["config_record"]), v["width"], v["height"], v["pixfmt"], v["line_bytes"], v["flags"])
    dec = api.Ffv1Decoder(v["width"], v["height"], v["pixfmt"], v["line_bytes"], cfg.num_h_slices, cfg.num_v_slices, cfg.slicecrc, cfg.context,
                          max_batch=len(packets), flags=v["flags"], coder=cfg.coder)
    assert dec.decode_host(packets, len(payloads[0])) == payloads
    with pytest.raises(RuntimeError):
        dec.decode_host([packets[0][:-9] + bytes([packets[0][-9] ^ 1]) + packets[0][-8:]], len(payloads[0]))
    dec.close()


def test_md5_of_host_buffers(built):
    """rcgpu_md5_host_batch == hashlib for sizes around the 64-byte block and padding boundaries (RFC 1321), incl. empty."""
    import numpy as np
    rng = np.random.default_rng(3)
    bufs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 65536, 1 << 20)]
    assert api.md5_host_batch(bufs) == [hashlib.md5(b).digest() for b in bufs]


DPX_LAYOUTS = [p for p in range(23) if p != synth.PIX_EXR_RGB16]


def test_analysis_of_a_batch_of_files(built):
    """rcgpu_analysis_host_batch: whole files of several layouts and sizes in one call -> hashlib's MD5 of every file and, for the DPX
    files, the padding verdict of oracle/dpx_oracle.c (DPX.cpp:501-608); a WAV file and junk are hashed and not scanned."""
    import numpy as np
    rng = np.random.default_rng(5)
    files, want = [], []
    for k, (pixfmt, w, h) in enumerate([(synth.PIX_RGB10_FILLEDA_BE, 37, 9), (synth.PIX_RGB10_FILLEDA_BE, 37, 9), (synth.PIX_RGB10_FILLEDA_BE, 37, 9),
                                        (synth.PIX_RGB16_BE, 21, 6), (synth.PIX_RGB12_PACKED_BE, 50, 7), (synth.PIX_Y10_FILLEDA_BE, 64, 5),
                                        (synth.PIX_RGBA12_FILLEDA_LE, 16, 4), (synth.PIX_RGB10_FILLEDA_BE, 40, 9)]):
        bits, nc, _, be = synth.PIX_INFO[pixfmt]
        packing = synth.DPX_PACKING.get(pixfmt, 1 if bits in (10, 12) else 0)
        payload, _ = synth.pack_payload(synth.components(w, h, nc, bits, "noise", seed=k), pixfmt, True)
        payload = bytearray(payload)
        if k in (1, 4, 6):                                                  # a few bits anywhere: some land in padding, some do not
            for _ in range(3):
                payload[int(rng.integers(0, len(payload)))] ^= 1 << int(rng.integers(0, 8))
        if k == 2:
            payload[len(payload) - 1] |= 1
        files.append(synth.dpx_file(None, pixfmt, frame_index=k, payload=bytes(payload), size=(w, h)))
        first = ob.dpx_padding_first_nonzero(bytes(payload), w, h, bits, nc, be, packing, False)
        want.append((hashlib.md5(files[-1]).digest(), True, -1 if first == 2 ** 64 - 1 else first))
    files.append(synth.wav_file(synth.pcm_samples(500, 2, 16), 16)); want.append((hashlib.md5(files[-1]).digest(), False, -1))
    files.append(os.urandom(5000)); want.append((hashlib.md5(files[-1]).digest(), False, -1))
    files.append(files[0][:3000]); want.append((hashlib.md5(files[-1]).digest(), False, -1))      # a truncated DPX file is not scanned
    assert api.analysis_host_batch(files) == want
    assert any(w[2] >= 0 for w in want) and any(w[1] and w[2] < 0 for w in want)


@pytest.mark.parametrize("pixfmt", DPX_LAYOUTS)
def test_padding_scan_matches_the_reference_rule(built, pixfmt):
    """rcgpu_dpx_padding_scan_device against oracle/dpx_oracle.c (DPX.cpp:501-608 restated from the header facts): clean payloads,
    then payloads with single bits set at random places -- inside samples (must be ignored unless the reference's own mask covers
    them), in filler bits, in line padding, in the last word."""
    import numpy as np
    bits, nc, _, be = synth.PIX_INFO[pixfmt]
    packing = synth.DPX_PACKING.get(pixfmt, 1 if bits in (10, 12) else 0)
    rng = np.random.default_rng(100 + pixfmt)
    for (w, h) in ((37, 9), (64, 5), (50, 7), (3, 2)):
        for altern in ((False, True) if pixfmt in (synth.PIX_Y10_FILLEDA_BE, synth.PIX_Y10_FILLEDB_BE) else (False,)):
            flags = synth.FLAG_ALTERN if altern else 0
            clean, _ = synth.pack_payload(synth.components(w, h, nc, bits, "noise", seed=w + h), pixfmt, True, flags)
            variants = [bytes(clean)]
            for k in range(40):
                b = bytearray(clean)
                for _ in range(int(rng.integers(1, 4))):
                    at = int(rng.integers(0, len(b))) if k % 3 else len(b) - 1 - int(rng.integers(0, min(len(b), 8)))
                    b[at] ^= 1 << int(rng.integers(0, 8))
                variants.append(bytes(b))
            zeroed = bytearray(len(clean))                                   # all-zero payload, then one bit anywhere
            variants.append(bytes(zeroed))
            for k in range(40):
                b = bytearray(len(clean)); b[int(rng.integers(0, len(b)))] = 1 << int(rng.integers(0, 8))
                variants.append(bytes(b))
            want = [ob.dpx_padding_first_nonzero(v, w, h, bits, nc, be, packing, altern) for v in variants]
            dv = [dev(v) for v in variants]
            got = api.dpx_padding_scan_device([t.data_ptr() for t in dv], pixfmt, w, h, flags)
            assert got == want, (pixfmt, w, h, altern, [i for i in range(len(want)) if got[i] != want[i]][:5])
            assert want[0] == 2 ** 64 - 1 or packing == 0                     # synth writes zero filler bits
    with pytest.raises(RuntimeError):
        api.dpx_padding_scan_device([0], synth.PIX_EXR_RGB16, 8, 8)


def test_padding_scan_matches_the_reference_parser(built):
    """rcgpu_dpx_padding_scan_device against tests/golden/padding_vectors.json: the reference's own In_FirstNonZero for 684 files
    (dpx::ParseBuffer, DPX.cpp:501-608, run by oracle/ref_padding_probe.cpp; recipe in tests/golden/make_padding_golden.py)."""
    import json
    from test_oracle import _padding_payload, _padding_vectors
    vs = _padding_vectors()
    groups = {}
    for v in vs:
        groups.setdefault((v["pixfmt"], v["width"], v["height"], v["flags"]), []).append(v)
    checked = 0
    for (pixfmt, w, h, flags), items in groups.items():
        dv = [dev(_padding_payload(v)) for v in items]
        got = api.dpx_padding_scan_device([t.data_ptr() for t in dv], pixfmt, w, h, flags)
        want = [2 ** 64 - 1 if v["first_nonzero"] < 0 else v["first_nonzero"] for v in items]
        assert got == want, (items[0]["flavor"], w, h, flags, [i for i in range(len(want)) if got[i] != want[i]][:5])
        checked += len(items)
    assert checked == len(vs) > 600


@pytest.mark.parametrize("seed", range(int(os.environ.get("RCGPU_SOAK_EXT", "24"))))
def test_random_streams_of_the_general_syntax(built, seed):
    """Random stream descriptions (table sets, set per plane group, transitions, coded initial states, version 1 headers; tests/ext_streams.py)
    written by the oracle and decoded by the device knowing only the geometry: the payload bytes, with the full window and with samples forced
    down the careful path -- and by the REAL reference's decoder too where its driver was built (oracle/_ref/ref_ffv1_decode: the same bytes
    from ffv1_frame::Process), so that every one of these streams is a stream the reference reads the same way.  RCGPU_SOAK_EXT=N runs N seeds."""
    import ext_streams
    import ref_decode
    w, h, pixfmt, line_bytes, rec, pks, pls, flavor, flags, tight = ext_streams.random_stream(seed)
    code, back = ob.decode_stream(ob.Params(w, h, pixfmt), rec, pks[0], line_bytes)
    assert code == 0 and back == pls[0]
    if ref_decode.available():
        (frames,), lines = ref_decode.decode([(flavor, flags, w, h, rec, pks)])
        assert ext_streams.reference_decodes_to(frames, tight), lines
    stream = api.Ffv1Stream(rec, pks[0])
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, max_batch=len(pks), stream=stream)
    for cap in (7, 1 + seed % 3):
        dec.debug_window(cap)
        assert dec.decode_host(pks, len(pls[0])) == pls, f"window {cap}"
    dec.close()


def test_a_hinted_file_that_changed_is_mapped_anew(built, tmp_path):
    """rcgpu_ffv1_decoder_decode_keep_hint_file maps the Matroska file for itself and keeps the mapping -- while it is still THAT file: a file
    rewritten under the same name (other size, other bytes, another inode) is mapped again instead of being read through the stale mapping;
    and a hinted batch that fails says why when it is adopted."""
    w, h, pixfmt, nh, nv = 192, 96, synth.PIX_RGB16_BE, 2, 2
    sets = []
    for k in range(2):
        srcs = [synth.pack_payload(synth.components(w, h, 3, 16, "film", seed=300 + 10 * k + i), pixfmt, True) for i in range(3)]
        line_bytes = srcs[0][1]
        sets.append([s_[0] for s_ in srcs])
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3)
    packets = [enc.encode_host(s_) for s_ in sets]
    enc.close()
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=3)
    path = str(tmp_path / "blocks.bin")

    def write(pk, pad):
        offs, pos, blob = [], pad, bytearray(pad)
        for p_ in pk:
            offs.append(pos); blob += p_ + b"\0" * 7; pos += len(p_) + 7
        tmp = path + ".new"
        open(tmp, "wb").write(blob)
        os.replace(tmp, path)                                                                 # another inode under the same name
        return offs
    offs = write(packets[0], 11)
    dec.decode_keep_hint_file(path, offs, [len(p_) for p_ in packets[0]])
    dec.decode_keep_adopt()
    assert [dec.kept_to_host(i, len(sets[0][i])) for i in range(3)] == sets[0]
    offs = write(packets[1], 4099)                                                              # other bytes at other places, another size
    dec.decode_keep_hint_file(path, offs, [len(p_) for p_ in packets[1]])
    dec.decode_keep_adopt()
    assert [dec.kept_to_host(i, len(sets[1][i])) for i in range(3)] == sets[1]
    with pytest.raises(api.RcgpuError, match="ends before a packet"):
        dec.decode_keep_hint_file(path, [offs[0], os.path.getsize(path) - 5, offs[2]], [len(p_) for p_ in packets[1]])
    dec.decode_keep_hint_file(path, [offs[0] + 3, offs[1], offs[2]], [len(p_) for p_ in packets[1]])       # a batch that cannot be decoded
    with pytest.raises(api.RcgpuError, match="decoded ahead failed: .*undecodable"):
        dec.decode_keep_adopt()
    dec.close()


def test_a_device_without_room_is_an_error_not_a_crash(built):
    """Out of device memory is an ordinary error of rcgpu_ffv1_decoder_create / rcgpu_ffv1_create (100 / "device setup failed"): route C halves its
    batch on it (oracle/route_c_ffv1_frame_cpp.patch), the pipeline sizes by free memory.  Also after decoders have come and gone in the
    process -- the order in which the HIP runtime of ROCm 7.0 used to die (CU-masked streams are pooled since, ffv1_check.hip)."""
    w, h, pixfmt = 1024, 540, synth.PIX_RGB16_BE
    line_bytes = w * 6
    for _ in range(2):
        d = api.Ffv1Decoder(w, h, pixfmt, line_bytes, 4, 4, 1, 1, max_batch=2)
        d.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    filler = torch.empty(free0 - (256 << 20), dtype=torch.uint8, device="cuda")
    with pytest.raises(api.RcgpuError, match="device setup failed") as ei:
        api.Ffv1Decoder(w, h, pixfmt, line_bytes, 4, 4, 1, 1, max_batch=512)                 # 512 x 16 chains x 10126 contexts x 32 bytes = 2.6 GB of states
    assert ei.value.code == 100
    with pytest.raises(api.RcgpuError):
        api.Ffv1Encoder(w, h, pixfmt, line_bytes, 4, 4, 1, 1, max_batch=512)
    small = api.Ffv1Decoder(w, h, pixfmt, line_bytes, 4, 4, 1, 1, max_batch=1)              # what a halving caller arrives at
    small.close()
    del filler
    torch.cuda.empty_cache()
    big = api.Ffv1Decoder(w, h, pixfmt, line_bytes, 4, 4, 1, 1, max_batch=512)
    big.close()
