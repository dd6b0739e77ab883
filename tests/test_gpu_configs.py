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


def test_config2_4k_batches_run_on(built):
    """The headline shape in the encoder's run-on mode with DIFFERENT pictures in every batch (the bench codes the same ones again and again:
    a batch that found another batch's symbols, states or byte buffers would not show there): four batches of six 4K frames, batch k+1
    issued while batch k is coded; every packet decodes to ITS source on the device, and equals what the same encoder produces one batch at
    a time."""
    w, h, pixfmt, slices, n, nb = 4096, 2160, synth.PIX_RGB16_BE, 64, 6, 4
    nh, nv = api.slices_to_grid(slices)
    srcs = []
    for i in range(n * nb):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 16, "film" if i % 5 else "noise", seed=900 + i), pixfmt, True)
        srcs.append(pl)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n, rc_span=1)
    dsrc = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in srcs]
    stride = (enc.max_packet + 255) & ~255
    st = torch.cuda.Stream()

    def run(run_on):
        dpk = [torch.zeros(n * stride, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        dsz = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(nb)]
        torch.cuda.synchronize()
        enc.set_run_on(run_on)
        for b in range(nb):
            enc.encode_device([t.data_ptr() for t in dsrc[b * n:(b + 1) * n]], dpk[b].data_ptr(), stride, dsz[b].data_ptr(), st.cuda_stream)
        enc.join(st.cuda_stream); st.synchronize()
        assert enc.error_flags() == 0
        return dpk, [z.cpu().tolist() for z in dsz]
    pk1, sz1 = run(True)
    pk0, sz0 = run(False)
    assert sz1 == sz0
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=n)
    dout = [torch.empty(len(srcs[0]), dtype=torch.uint8, device="cuda") for _ in range(n)]
    for b in range(nb):
        for i in range(n):
            assert torch.equal(pk1[b][i * stride:i * stride + sz1[b][i]], pk0[b][i * stride:i * stride + sz0[b][i]]), (b, i)
        assert dec.decode_device([pk1[b].data_ptr() + i * stride for i in range(n)], sz1[b], [t.data_ptr() for t in dout]) == 0
        for i in range(n):
            assert api.compare_device(dout[i].data_ptr(), dsrc[b * n + i].data_ptr(), len(srcs[0])) == -1, (b, i)
    enc.close(); dec.close()


def test_a_4k_slice_of_the_device_decodes_by_the_rfc_alone(built):
    """One 512x270 slice of a 4K frame the device coded (config 2's shape: 64 slices, 5063 contexts, 17-bit differences), decoded by
    tests/rfc9043_validator.py -- code written from RFC 9043's text, none of this repository's codec -- gives the source's pixels: the
    sample-level conformance evidence for the GPU bitstream that does not go through the reference's decoder or its restatement."""
    import numpy as np
    import rfc9043_validator as rfc
    w, h, pixfmt = 4096, 2160, synth.PIX_RGB16_BE
    comp = synth.components(w, h, 3, 16, "film", seed=77)
    pl, line_bytes = synth.pack_payload(comp, pixfmt, True)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 8, 8, 1, 1, max_batch=1)
    packet = enc.encode_host([pl])[0]
    record = enc.config_record()
    enc.close()
    r = rfc.parse_record(record)
    rfc.validate_frame(r, packet, w, h)
    k = 27                                                     # slice (3, 3)
    px = rfc.decode_frame(r, packet, w, h, only={k})
    y0, x0 = 3 * 270, 3 * 512
    got = np.array([row[x0:x0 + 512] for row in px[y0:y0 + 270]], dtype=np.int64)
    assert got.shape == (270, 512, 3) and np.array_equal(got, comp[y0:y0 + 270, x0:x0 + 512].astype(np.int64))


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


# ---- whole jobs at the configs' full picture size: files -> rcgpu-ffmpeg (GPU) -> MKV -> the REAL reference's --check ----
import os          # noqa: E402
import shlex       # noqa: E402
import shutil      # noqa: E402
import subprocess  # noqa: E402
import tempfile    # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
OK_LINE = "Reversibility was checked, no issue detected."


def _run(cmd, cwd, timeout, attempts=2, env=None):
    for a in range(attempts):                      # one retry: the lost wake-up of the reference's thread pool (ThreadPool.h:27-33,66-68; tests/test_gpu_e2e.py::run, profiles/r04_hang_stacks.txt)
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL, env=env)
        except subprocess.TimeoutExpired:
            if a + 1 == attempts:
                raise


def _film_payloads(n, w, h):
    """n synthetic RGB16 big-endian 'film' payloads, made on the device (numpy needs seconds per 4K frame)."""
    import bench
    return bench.make_frames(torch, n, w, h, "film", 7, torch.device("cuda", 0)).cpu().numpy()


def _job(work, n_expected_slices, extra_check=None):
    ref = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/rawcooked not built (needs /root/reference)")
    r = _run([ref, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work, 300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("-slices %d " % n_expected_slices) in r.stdout, r.stdout
    argv = shlex.split(r.stdout.strip())
    r = _run([SHIM] + argv[1:], work, 300, env=dict(os.environ, RCGPU_TRACE="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    import mkv_validator
    rep = mkv_validator.validate(os.path.join(work, "pkg.mkv"))
    r2 = _run([ref, "--check", "pkg.mkv"], work, 900)
    assert r2.returncode == 0 and OK_LINE in r2.stdout and "Error" not in (r2.stdout + r2.stderr), r2.stdout + r2.stderr
    return rep, r.stderr


def test_config3_job_4k_dpx_with_6ch_wav_through_the_reference(built):
    """BASELINE config 3 as a job: 4K-DCI 16-bit DPX + 6 ch / 24 bit / 48 kHz WAV -> FFV1 + FLAC in one MKV, coded on the GPU from the
    command the reference prints, several batches (RCGPU_BATCH), then rebuilt and hash-checked by the reference's CPU decoders."""
    n, w, h = 20, 4096, 2160
    work = tempfile.mkdtemp(prefix="rcgpu_cfg3_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        os.makedirs(os.path.join(work, "pkg", "img"))
        frames = _film_payloads(n, w, h)
        for i in range(n):
            with open(os.path.join(work, "pkg", "img", "f_%06d.dpx" % i), "wb") as f:
                f.write(synth.dpx_file(None, synth.PIX_RGB16_BE, frame_index=i, payload=frames[i].tobytes(), size=(w, h)))
        with open(os.path.join(work, "pkg", "snd.wav"), "wb") as f:
            f.write(synth.wav_file(synth.pcm_samples(n * 2000, 6, 24, 48000), 24, 48000))        # n / 24 s of audio
        os.environ["RCGPU_BATCH"] = "8"
        try:
            rep, trace = _job(work, 576)                                                          # the reference's own choice for 4K 16 bit
        finally:
            del os.environ["RCGPU_BATCH"]
        assert rep["tracks"][1] == {"type": 1, "codec": "V_FFV1", "blocks": n} and rep["tracks"][2]["codec"] == "A_FLAC" and rep["tracks"][2]["blocks"] >= 8
        assert "batches of 8" in trace
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_config4_job_8k_tiff_sequence_through_the_reference(built):
    """BASELINE config 4's shape as a job: 8 frames of 8192x4320 16-bit little-endian TIFF, the reference's default 576 slices, batches
    of 3 (several batches in flight, a ragged last one), rebuilt and hash-checked by the reference."""
    n, w, h = 8, 8192, 4320
    work = tempfile.mkdtemp(prefix="rcgpu_cfg4_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        os.makedirs(os.path.join(work, "pkg", "img"))
        be = _film_payloads(n, w, h)
        import struct
        import numpy as np
        # header and IFD of a small TIFF from synth, patched to w x h; the payload follows them
        small = synth.tiff_file(np.zeros((2, 2, 3), dtype=np.uint16), synth.PIX_RGB16_LE)
        head = bytearray(small[:api.tiff_probe(small).data_offset])
        for t in range(struct.unpack_from("<H", head, 8)[0]):
            at = 10 + 12 * t
            tag = struct.unpack_from("<H", head, at)[0]
            if tag == 256:
                struct.pack_into("<I", head, at + 8, w)
            if tag in (257, 278):
                struct.pack_into("<I", head, at + 8, h)
            if tag == 279:
                struct.pack_into("<I", head, at + 8, w * h * 6)
        for i in range(n):
            with open(os.path.join(work, "pkg", "img", "f_%06d.tif" % i), "wb") as f:
                f.write(bytes(head)); f.write(be[i].reshape(-1, 2)[:, ::-1].tobytes())           # the same samples, little endian
        os.environ["RCGPU_BATCH"] = "3"
        try:
            rep, trace = _job(work, 576)
        finally:
            del os.environ["RCGPU_BATCH"]
        assert rep["tracks"][1]["blocks"] == n and "batches of 3" in trace
    finally:
        shutil.rmtree(work, ignore_errors=True)
