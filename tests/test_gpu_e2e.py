"""End to end on the GPU box, judged by the REAL reference: rawcooked (oracle/_ref) analyses the sources and prints the
encoder command; our ffmpeg-argv shim (rcgpu-ffmpeg -> librcgpu.so -> MI355X) executes it; `rawcooked --check` must then
rebuild every source file bit-exactly (this mirrors Project/GNU/CLI/test/test1.sh:22-67 and vulkan.sh:48-80 of the
reference, with the encoder swapped)."""
import os
import shlex
import shutil
import subprocess
import sys

import numpy as np
import pytest

from rawcooked_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
OK_LINE = "Reversibility was checked, no issue detected."      # test2.sh:37,69


def make_package(work, w, h, pixfmt, nframes, kind, tiff=False, audio=None, start=0, flags=0, exr=False):
    os.makedirs(os.path.join(work, "pkg", "img"))
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    for i in range(nframes):
        comp = synth.components(w, h, nc, bits, kind, seed=7 * i + 1)
        data = synth.exr_file(comp) if exr else synth.tiff_file(comp, pixfmt, trailer=b"tail123") if tiff else synth.dpx_file(comp, pixfmt, frame_index=i, flags=flags)
        with open(os.path.join(work, "pkg", "img", "f_%06d.%s" % (start + i, "exr" if exr else "tif" if tiff else "dpx")), "wb") as f:
            f.write(data)
    if audio:
        ch, abits, rate, n = audio
        with open(os.path.join(work, "pkg", "snd.wav"), "wb") as f:
            f.write(synth.wav_file(synth.pcm_samples(n, ch, abits, rate), abits, rate))


def run(cmd, cwd, timeout=20, attempts=2):
    """Run one step of the round trip.  ONE retry, for one known cause in unpatched reference code: the lost wake-up of its third-party
    thread pool -- `shutdown()` sets `m_shutdown` and calls `notify_all()` without the mutex (Lib/ThirdParty/thread-pool/include/
    ThreadPool.h:66-68) while a worker that has just tested `while (!m_pool->m_shutdown)` has not reached its `wait()` yet (:27-33); the
    worker then sleeps for good and `matroska::Shutdown()` (Lib/Compressed/Matroska/Matroska.cpp:271-274, `FramesPool->shutdown()`) never
    returns from `std::thread::join`.  tools/hang/hang_hunt.py caught it six times in 750 runs on the GPU box (profiles/r04_hang_stacks.txt:
    every one of them this picture -- main thread in Shutdown() -> join, one pool worker in condition_variable::wait, the HSA runtime's two
    event threads idle, no thread inside rcgpu_* or an RCGPU_LINKED block); every step is idempotent (-y)."""
    for attempt in range(attempts):
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL)
        except subprocess.TimeoutExpired:
            sys.stderr.write("e2e: step timed out after %d s (attempt %d): %s\n" % (timeout, attempt + 1, " ".join(cmd[:6])))
            if os.environ.get("RCGPU_E2E_LOG"):
                with open(os.environ["RCGPU_E2E_LOG"], "a") as f:
                    f.write("timeout attempt %d: %s\n" % (attempt + 1, " ".join(cmd)))
            if attempt + 1 == attempts:
                raise


CASES = [
    dict(w=64, h=48, pixfmt=synth.PIX_RGB16_BE, nframes=3, kind="film"),
    dict(w=50, h=38, pixfmt=synth.PIX_RGB10_FILLEDA_BE, nframes=2, kind="film"),
    dict(w=61, h=35, pixfmt=synth.PIX_RGB10_FILLEDA_LE, nframes=2, kind="noise"),
    dict(w=40, h=30, pixfmt=synth.PIX_RGB16_LE, nframes=2, kind="film", tiff=True),
    dict(w=48, h=32, pixfmt=synth.PIX_RGB8, nframes=2, kind="film"),
    dict(w=34, h=17, pixfmt=synth.PIX_RGB12_FILLEDA_BE, nframes=2, kind="film"),   # even width: the reference itself mis-sizes odd-width 12-bit FilledA (DPX.cpp:470-474 vs RawFrame.cpp:109)
    dict(w=46, h=21, pixfmt=synth.PIX_RGB12_FILLEDA_LE, nframes=2, kind="noise"),
    dict(w=32, h=24, pixfmt=synth.PIX_RGBA16_BE, nframes=2, kind="film"),
    dict(w=40, h=24, pixfmt=synth.PIX_Y16_BE, nframes=2, kind="film"),
    dict(w=40, h=24, pixfmt=synth.PIX_Y8, nframes=2, kind="film"),
    dict(w=64, h=48, pixfmt=synth.PIX_RGB16_BE, nframes=4, kind="film", audio=(6, 24, 48000, 9000)),   # config 3 shape
    dict(w=2048, h=1556, pixfmt=synth.PIX_RGB10_FILLEDA_BE, nframes=2, kind="film"),                    # config 1 shape
    # bit-packed flavors (geometries: see the note in tests/golden/make_golden.py about the reference's own word merge)
    dict(w=56, h=38, pixfmt=synth.PIX_RGB12_PACKED_BE, nframes=2, kind="film"),
    dict(w=96, h=40, pixfmt=synth.PIX_RGB12_PACKED_BE, nframes=2, kind="film", flags=synth.FLAG_VFLIP),    # DPX orientation 2 -> "-vf vflip"
    dict(w=96, h=40, pixfmt=synth.PIX_RGBA10_FILLEDA_BE, nframes=2, kind="film"),
    dict(w=51, h=38, pixfmt=synth.PIX_RGBA10_FILLEDA_LE, nframes=2, kind="noise"),
    dict(w=50, h=38, pixfmt=synth.PIX_RGBA12_PACKED_BE, nframes=2, kind="film"),
    dict(w=50, h=38, pixfmt=synth.PIX_RGBA12_FILLEDA_BE, nframes=2, kind="film"),
    dict(w=52, h=38, pixfmt=synth.PIX_Y10_FILLEDA_BE, nframes=2, kind="film"),
    dict(w=50, h=38, pixfmt=synth.PIX_Y10_FILLEDB_BE, nframes=2, kind="film", flags=synth.FLAG_ALTERN),     # words run across line ends
    dict(w=56, h=38, pixfmt=synth.PIX_Y12_PACKED_BE, nframes=2, kind="film"),
    dict(w=96, h=40, pixfmt=synth.PIX_Y12_PACKED_BE, nframes=2, kind="noise", flags=synth.FLAG_VFLIP),
    dict(w=72, h=40, pixfmt=synth.PIX_EXR_RGB16, nframes=3, kind="film", exr=True),                         # "-c:v exr -consider_float16_as_uint16 1"
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d-%d%s%s%s" % (c["w"], c["h"], c["pixfmt"], "-tiff" if c.get("tiff") else "", "-wav" if c.get("audio") else "", "-f%d" % c["flags"] if c.get("flags") else ""))
def test_reference_accepts_gpu_mkv(built, refbin, tmp_path, case):
    work = str(tmp_path)
    make_package(work, case["w"], case["h"], case["pixfmt"], case["nframes"], case["kind"], case.get("tiff", False), case.get("audio"), flags=case.get("flags", 0), exr=case.get("exr", False))
    # 1. the reference analyses the package, writes the reversibility data and prints the ffmpeg command (-d)
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"] + (["--check"] if case.get("exr") else []), work)   # EXR: Main.cpp:121-127
    assert r.returncode == 0, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    assert argv[0] == "ffmpeg"
    # 2. our shim executes exactly that command line on the GPU
    r = run([SHIM] + argv[1:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(work, "pkg.mkv"))
    if case.get("exr"):                                                                     # Output.cpp:123 -metadata:s:v WARNING=... becomes a track tag
        assert "-metadata:s:v" in argv and b"Pixel content is IEEE 754 floating-point format" in open(os.path.join(work, "pkg.mkv"), "rb").read(4096)
    # 3. the reference decodes the MKV and compares every rebuilt file with the hashes it stored
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    # 4. full decode, byte compare
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(work, "pkg")):
        for fn in files:
            src = os.path.join(dirpath, fn)
            dst = os.path.join(work, "pkg.mkv.RAWcooked", os.path.relpath(src, work))
            assert open(src, "rb").read() == open(dst, "rb").read(), fn


def test_unmodified_rawcooked_drives_the_shim(built, refbin, tmp_path):
    """rawcooked --bin-name <shim>: the reference itself launches our encoder through system() (Output.cpp:356)
    and then runs its own post-encode check."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 3, "film", audio=(2, 16, 48000, 6000))
    r = run([refbin, "--bin-name", SHIM, "--check", "--hash", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    assert os.path.exists(os.path.join(work, "pkg.mkv"))


def test_unmodified_rawcooked_selects_the_gpu_the_reference_s_way(built, refbin, tmp_path):
    """`rawcooked -c:v ffv1_vulkan[:N]` is the reference's own GPU-selection surface (CLI/Global.cpp:367-378: -init_hw_device "vulkan=vk:N" -vf hwupload
    -c:v ffv1_vulkan): a site that runs it with --bin-name gets this encoder on HIP device N.  Mirrors test/vulkan.sh:48-80 -- encode, decode, byte
    compare -- and a device that is not there is an Error: line and a failed job, never another device."""
    import torch
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 3, "film", audio=(2, 16, 48000, 6000))
    r = run([refbin, "--bin-name", SHIM, "-c:v", "ffv1_vulkan:0", "--check", "--hash", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(work, "pkg")):
        for fn in files:
            src = os.path.join(dirpath, fn)
            assert open(src, "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", os.path.relpath(src, work)), "rb").read(), fn
    # the stream is what `-c:v ffv1_vulkan` means to FFmpeg without -context: the small context model (test/vulkan.sh:54 takes -context 1 out)
    n = torch.cuda.device_count()
    os.remove(os.path.join(work, "pkg.mkv"))
    r = run([refbin, "--bin-name", SHIM, "-c:v", "ffv1_vulkan:%d" % n, "--check", "--hash", "-y", "pkg"], work)
    assert r.returncode != 0 and "outside the %d visible devices" % n in (r.stdout + r.stderr), r.stdout + r.stderr
    assert not os.path.exists(os.path.join(work, "pkg.mkv")) or os.path.getsize(os.path.join(work, "pkg.mkv")) == 0


def test_output_version_2_appends_to_our_mkv(built, refbin, tmp_path):
    """--output-version 2 (Main.cpp:905-929): the encoder gets no reversibility attachment; rawcooked appends its own EBML document
    behind our Segment afterwards -- which only works because the Segment's size is exact -- and then reads the file back."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB10_FILLEDA_BE, 3, "film", audio=(2, 16, 48000, 6000))
    r = run([refbin, "--output-version", "2", "--bin-name", SHIM, "--check", "--hash", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    mkv = open(os.path.join(work, "pkg.mkv"), "rb").read()
    assert mkv.count(b"\x1a\x45\xdf\xa3") >= 2                                           # a second EBML header follows the Segment
    r = run([refbin, "-y", "pkg.mkv"], work)                                              # and the files come back byte for byte
    assert r.returncode == 0, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(work, "pkg")):
        for fn in files:
            src = os.path.join(dirpath, fn)
            assert open(src, "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", os.path.relpath(src, work)), "rb").read(), fn


def test_non_zero_padding_bits_survive_through_the_reversibility_data(built, refbin, tmp_path):
    """--check-padding: the reference's analysis finds filler bits that are not zero (DPX.cpp:501-608) and keeps them in the
    reversibility data; the encoder must ignore them and the rebuilt files must carry them again."""
    work = str(tmp_path)
    make_package(work, 50, 38, synth.PIX_RGB10_FILLEDA_BE, 3, "film")
    for i in range(3):
        p = os.path.join(work, "pkg", "img", "f_%06d.dpx" % i)
        d = bytearray(open(p, "rb").read())
        off = int.from_bytes(d[4:8], "big")                               # offset to image data
        for k in range(off + 3, len(d), 4 * (7 + i)):                     # low two bits of some big-endian words
            d[k] |= 1 + (k % 3 == 0)
        open(p, "wb").write(d)
    r = run([refbin, "--bin-name", SHIM, "--check-padding", "--check", "--hash", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(3):
        rel = os.path.join("pkg", "img", "f_%06d.dpx" % i)
        assert open(os.path.join(work, rel), "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", rel), "rb").read()


def test_flipped_byte_is_detected(built, refbin, tmp_path):
    """Negative control (paddingbits.sh / check.sh style): corrupt one slice byte -> the reference must object."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 2, "film")
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    argv = shlex.split(r.stdout.strip())
    assert run([SHIM] + argv[1:], work).returncode == 0
    p = os.path.join(work, "pkg.mkv")
    data = bytearray(open(p, "rb").read())
    data[len(data) - 2000] ^= 0x40
    open(p, "wb").write(data)
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode != 0 or "Error" in (r.stdout + r.stderr) or OK_LINE not in r.stdout


def test_reference_accepts_compact_context_model(built, refbin, tmp_path):
    """`-rcgpu_context_model compact` (an output option of the shim, rcgpu_job::options): custom quantisation tables travel in the configuration record; the reference's decoder must
    rebuild the sources from them (FFV1_Parameters.cpp:206-253 parses any monotone table)."""
    work = str(tmp_path)
    make_package(work, 160, 90, synth.PIX_RGB16_BE, 3, "film", audio=(2, 24, 48000, 5000))
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    at = argv.index("-coder")          # among the output options (CLI/Output.cpp:273-278)
    r = run([SHIM] + argv[1:at] + ["-rcgpu_context_model", "compact"] + argv[at:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    compact_size = os.path.getsize(os.path.join(work, "pkg.mkv"))
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    assert run([SHIM] + argv[1:], work).returncode == 0          # FFmpeg's level maps, for the size comparison
    assert compact_size < os.path.getsize(os.path.join(work, "pkg.mkv"))


def test_start_number_sidecar_and_no_overwrite(built, refbin, tmp_path):
    """Non-zero -start_number (Input.cpp:305-306 template), an extra attachment (Output.cpp:279-287) and -n (Global.cpp:865-884)."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 3, "film", start=86400)
    with open(os.path.join(work, "pkg", "notes.txt"), "w") as f:
        f.write("sidecar kept as attachment\n")
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0 and "-start_number 086400" in r.stdout and "notes.txt" in r.stdout, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    assert run([SHIM] + argv[1:], work).returncode == 0
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert open(os.path.join(work, "pkg.mkv.RAWcooked", "pkg", "notes.txt")).read() == "sidecar kept as attachment\n"
    # -n: refuse to overwrite
    argv_n = [a if a != "-y" else "-n" for a in argv[1:]]
    r = run([SHIM] + argv_n, work)
    assert r.returncode != 0 and "Error: " in r.stderr and "already exists" in r.stderr


@pytest.mark.parametrize("rate", [None, "25"], ids=["file-header-rate", "25fps-bare-paths"])
def test_gapped_sequence_uses_the_concat_list(built, refbin, tmp_path, rate):
    """--accept-gaps makes the reference write an ffconcat file list (Output.cpp:138-251); the shim must read it.  At exactly 25 frames
    per second the reference leaves the list as bare paths (Output.cpp:162-163) -- the shim reads those too."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 5, "film")
    os.remove(os.path.join(work, "pkg", "img", "f_000002.dpx"))
    r = run([refbin, "--hash", "--no-check-padding", "--accept-gaps", "-d", "-y"] + (["-framerate", rate] if rate else []) + ["pkg"], work)
    assert r.returncode == 0 and "-f concat" in r.stdout, r.stdout + r.stderr
    listing = open(os.path.join(work, [a for a in shlex.split(r.stdout.strip()) if a.endswith(".FileList.txt")][0])).read()
    assert ("file '" in listing) == (rate is None)
    argv = shlex.split(r.stdout.strip())
    r = run([SHIM] + argv[1:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr


def test_ntsc_framerate_and_python_job_binding(built, refbin, tmp_path):
    """The ctypes `Output.Process()` mirror of Output.h:41-53 with a 24000/1001 frame rate."""
    from rawcooked_amd import api
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB10_FILLEDA_BE, 4, "film")
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0
    cwd = os.getcwd()
    os.chdir(work)
    try:
        out = api.Output(Streams=[api.Stream(FileName_Template="pkg/img/f_%06d.dpx", FileName_StartNumber="000000",
                                             Flavor="DPX/Raw/RGB/10bit/U/BE/FilledA", Slices="16", FrameRate="24000/1001")])
        assert out.Process("pkg.mkv", rawcooked_reversibility_FileName="pkg.rawcooked_reversibility_data") == 0, api.last_error()
    finally:
        os.chdir(cwd)
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    data = open(os.path.join(work, "pkg.mkv"), "rb").read()
    i = data.index(bytes.fromhex("23E383"))                   # DefaultDuration
    assert int.from_bytes(data[i + 4:i + 4 + (data[i + 3] & 0x7F)], "big") == 41708333


def test_package_with_two_video_tracks_and_two_audio_tracks(built, refbin, tmp_path):
    """test2-style package: two image sequences of different flavors plus two WAV files -> four tracks in stream
    order (Output.cpp:258-268 `-map`), every rebuilt file identical."""
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "pkg", "a_16bit"))
    os.makedirs(os.path.join(work, "pkg", "b_12bit"))        # same slice count as the 16-bit track: the reference refuses mixed counts (one global -slices)
    for i in range(3):
        with open(os.path.join(work, "pkg", "a_16bit", "a_%04d.dpx" % i), "wb") as f:
            f.write(synth.dpx_file(synth.components(64, 48, 3, 16, "film", seed=i), synth.PIX_RGB16_BE, frame_index=i))
        with open(os.path.join(work, "pkg", "b_12bit", "b_%04d.dpx" % i), "wb") as f:
            f.write(synth.dpx_file(synth.components(64, 48, 3, 12, "film", seed=10 + i), synth.PIX_RGB12_FILLEDA_LE, frame_index=i))
    with open(os.path.join(work, "pkg", "c_stereo.wav"), "wb") as f:
        f.write(synth.wav_file(synth.pcm_samples(6000, 2, 24, 48000), 24, 48000))
    with open(os.path.join(work, "pkg", "d_mono.wav"), "wb") as f:
        f.write(synth.wav_file(synth.pcm_samples(6000, 1, 16, 48000, seed=3), 16, 48000))
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0 and r.stdout.count(" -i ") == 4 and "-map 3" in r.stdout, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    r = run([SHIM] + argv[1:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    n = 0
    for dirpath, _, files in os.walk(os.path.join(work, "pkg")):
        for fn in files:
            src = os.path.join(dirpath, fn)
            dst = os.path.join(work, "pkg.mkv.RAWcooked", os.path.relpath(src, work))
            assert open(src, "rb").read() == open(dst, "rb").read(), fn
            n += 1
    assert n == 8


def test_coder_2_carries_its_state_table(built, refbin, tmp_path):
    """`rawcooked -coder 2` (Global.cpp:380-393): range coder whose state-transition table travels in the configuration record as
    deltas to the default one (FFV1_Parameters.cpp:41-55); smaller packets than -coder 1 on the same content."""
    work = str(tmp_path)
    make_package(work, 160, 90, synth.PIX_RGB16_BE, 3, "film")
    sizes = {}
    for coder in ("1", "2"):
        r = run([refbin, "-coder", coder, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
        assert r.returncode == 0 and ("-coder %s " % coder) in r.stdout, r.stdout + r.stderr
        argv = shlex.split(r.stdout.strip())
        r = run([SHIM] + argv[1:], work)
        assert r.returncode == 0, r.stdout + r.stderr
        sizes[coder] = os.path.getsize(os.path.join(work, "pkg.mkv"))
        r = run([refbin, "--check", "pkg.mkv"], work)
        assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    assert sizes["2"] < sizes["1"]


def test_slices_1_means_ffv1_version_1(built, refbin, tmp_path):
    """`rawcooked -slices 1` turns into `-level 1 -slices 1` (Global.cpp:961-968): FFV1 version 1 frames, no CodecPrivate."""
    work = str(tmp_path)
    make_package(work, 96, 54, synth.PIX_RGB16_BE, 3, "film", audio=(2, 16, 48000, 6000))
    r = run([refbin, "-slices", "1", "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0 and "-level 1 " in r.stdout and "-slices 1 " in r.stdout, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    r = run([SHIM] + argv[1:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr


def test_flac_refuses_what_only_copy_can_carry(built, tmp_path):
    """32-bit WAV with `-c:a flac`: the reference stops before the encoder (CLI/Main.cpp:308-314); the shim says the same."""
    (tmp_path / "a.wav").write_bytes(synth.wav_file(synth.pcm_samples(5000, 2, 24), 32))
    r = run([SHIM, "-i", str(tmp_path / "a.wav"), "-c:a", "flac", "-y", "-f", "matroska", str(tmp_path / "o.mkv")], str(tmp_path))
    assert r.returncode != 0 and "Error: FLAC encoding is not supported with 32-bit audio input, use -c:a copy" in r.stderr, r.stderr
    assert not os.path.exists(tmp_path / "o.mkv")


def test_video_with_copied_32_bit_audio(built, refbin, tmp_path):
    """A package whose WAV is 32-bit: the reference switches the whole job to `-c:a copy` by itself; FFV1 beside a PCM track."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 3, "film")
    with open(os.path.join(work, "pkg", "snd.wav"), "wb") as f:
        f.write(synth.wav_file(synth.pcm_samples(6000, 2, 24), 32, float32=True))
    r = run([refbin, "--bin-name", SHIM, "--check", "-y", "pkg"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("opts", [["-slicecrc", "0"], ["-context", "0"], ["-slicecrc", "0", "-context", "0"]], ids=lambda o: "".join(o))
def test_slicecrc_0_and_context_0_through_the_reference(built, refbin, tmp_path, opts):
    """`rawcooked -slicecrc 0` / `-context 0` (Source/CLI/Global.cpp:337-485; accepted values Project/GNU/CLI/test/check.sh:87-98) at
    -level 3, multi-slice, with audio: the reference passes the options on, the shim codes them on the GPU, the reference decodes."""
    work = str(tmp_path)
    make_package(work, 160, 90, synth.PIX_RGB16_BE, 4, "film", audio=(2, 24, 48000, 5000))
    r = run([refbin] + opts + ["--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    for k in range(0, len(opts), 2):
        assert ("%s %s " % (opts[k], opts[k + 1])) in r.stdout, r.stdout
    assert "-level 3 " in r.stdout
    argv = shlex.split(r.stdout.strip())
    r = run([SHIM] + argv[1:], work)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([refbin, "--check", "pkg.mkv"], work)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    r = run([refbin, "-y", "pkg.mkv"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(4):
        rel = os.path.join("pkg", "img", "f_%06d.dpx" % i)
        assert open(os.path.join(work, rel), "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", rel), "rb").read()


@pytest.mark.parametrize("env", [{}, {"RCGPU_MKV_NO_MMAP": "1"}], ids=["mapped-output", "pwrite-output"])
def test_long_sequence_crosses_batches_and_ring_slots(built, refbin, tmp_path, env):
    """70 frames in batches of 16 with audio: several batches in flight (upload k+1, code k, download k-1), pinned slots and ring chunks
    reused, packets placed in frame order by parallel writers -- into the mapped file, or with pwrite where mapping is refused."""
    work = str(tmp_path)
    make_package(work, 96, 64, synth.PIX_RGB16_BE, 70, "film", audio=(2, 16, 48000, 140000))
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    argv = shlex.split(r.stdout.strip())
    r = subprocess.run([SHIM] + argv[1:] + [], cwd=work, capture_output=True, text=True, env=dict(os.environ, RCGPU_BATCH="16", **env), timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([refbin, "--check", "pkg.mkv"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr


def test_error_text_of_a_reader_thread_reaches_the_user(built, refbin, tmp_path):
    """A frame in the middle of the sequence differs in geometry: the message of the reader thread that met it is what the user sees."""
    work = str(tmp_path)
    make_package(work, 64, 48, synth.PIX_RGB16_BE, 12, "film")
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    argv = shlex.split(r.stdout.strip())
    bits, nc, _, _ = synth.PIX_INFO[synth.PIX_RGB16_BE]
    with open(os.path.join(work, "pkg", "img", "f_000007.dpx"), "wb") as f:
        f.write(synth.dpx_file(synth.components(80, 48, nc, bits, "film", seed=3), synth.PIX_RGB16_BE, frame_index=7))
    r = run([SHIM] + argv[1:], work)
    assert r.returncode != 0 and "Error: " in r.stderr and "f_000007.dpx differs in geometry" in r.stderr, r.stderr
    assert not os.path.exists(os.path.join(work, "pkg.mkv"))


@pytest.mark.parametrize("case", [dict(w=64, h=48, pixfmt=synth.PIX_RGB16_BE, audio=(2, 16, 48000, 6000)),
                                  dict(w=50, h=38, pixfmt=synth.PIX_RGB10_FILLEDA_BE, padding=True),
                                  dict(w=96, h=40, pixfmt=synth.PIX_RGB12_PACKED_BE, flags=synth.FLAG_VFLIP),
                                  dict(w=40, h=30, pixfmt=synth.PIX_RGB16_LE, tiff=True)],
                         ids=["rgb16+wav", "rgb10-nonzero-padding", "rgb12packed-vflip", "tiff16"])
def test_linked_reference_encodes_and_checks_on_the_device(built, linkedbin, refbin, tmp_path, case):
    """INTEGRATION.md routes B and C compiled: the reference built with oracle/route_b_output_cpp.patch (rcgpu_encode() in place of the
    system() call, Source/CLI/Output.cpp:354-375) and oracle/route_c_ffv1_frame_cpp.patch (the device decoder in place of the slice
    pool in ffv1_frame::Process, FFV1_Frame.cpp:134-228; raw_frame::Process then merges the `In` data, RawFrame.cpp:184-206) encodes
    AND checks in one run; the unmodified reference then checks the same file with its CPU decoder, and the linked binary checks a file
    it did not make."""
    work = str(tmp_path)
    make_package(work, case["w"], case["h"], case["pixfmt"], 3, "film", case.get("tiff", False), case.get("audio"), flags=case.get("flags", 0))
    if case.get("padding"):
        for i in range(3):
            p = os.path.join(work, "pkg", "img", "f_%06d.dpx" % i)
            d = bytearray(open(p, "rb").read())
            off = int.from_bytes(d[4:8], "big")
            for k in range(off + 3, len(d), 4 * (5 + i)):
                d[k] |= 1 + (k % 3 == 0)
            open(p, "wb").write(d)
    r = run([linkedbin, "--check-padding" if case.get("padding") else "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    assert "ffmpeg" not in r.stderr.lower()                                   # nothing was spawned
    r = run([refbin, "--check", "pkg.mkv"], work)                             # the CPU decoder agrees
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    r = run([linkedbin, "-y", "pkg.mkv"], work, timeout=60)                   # full decode through the device decoder, byte compare
    assert r.returncode == 0, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(work, "pkg")):
        for fn in files:
            src = os.path.join(dirpath, fn)
            assert open(src, "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", os.path.relpath(src, work)), "rb").read(), fn
    # a corrupted slice is refused by the device decoder behind the reference's own error path
    p = os.path.join(work, "pkg.mkv")
    data = bytearray(open(p, "rb").read())
    data[len(data) - 300] ^= 0x20
    open(p, "wb").write(data)
    r = run([linkedbin, "--check", "pkg.mkv"], work, timeout=60)
    assert r.returncode != 0 or "Error" in (r.stdout + r.stderr) or OK_LINE not in r.stdout


@pytest.mark.parametrize("batch", ["1", "5", "64"])
def test_linked_reference_checks_in_batches(built, linkedbin, refbin, tmp_path, monkeypatch, batch):
    """Route C batches: the demuxer (oracle/route_c_matroska_cpp.patch, Matroska.cpp:934-953) announces the blocks of the track that follow
    the one it hands over, ffv1_frame::Process (oracle/route_c_ffv1_frame_cpp.patch) decodes up to RCGPU_CHECK_BATCH of them in one device call
    and keeps the payloads until their turn.  Audio blocks lie between the video blocks; 13 frames do not divide by 5; one corrupt frame in
    the middle is reported as that frame's error and the frames around it still check."""
    work = str(tmp_path)
    make_package(work, 96, 64, synth.PIX_RGB16_BE, 13, "film", audio=(2, 16, 48000, 26000))
    monkeypatch.setenv("RCGPU_CHECK_BATCH", batch)
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)        # encode (route B) + check (route C)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    r = run([linkedbin, "-y", "pkg.mkv"], work, timeout=60)                                                # every file rebuilt through the batches
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(13):
        fn = os.path.join("pkg", "img", "f_%06d.dpx" % i)
        assert open(os.path.join(work, fn), "rb").read() == open(os.path.join(work, "pkg.mkv.RAWcooked", fn), "rb").read(), fn
    # flip a bit inside the 7th video block: that frame is undecodable, the verdict says so, nothing crashes
    import mkv_validator
    blocks = []
    mkv_validator.validate(os.path.join(work, "pkg.mkv"), on_block=lambda trk, t, a, b: blocks.append((trk, a, b)))
    video = [(a, b) for trk, a, b in blocks if trk == 1]
    assert len(video) == 13 and any(trk != 1 for trk, _, _ in blocks)
    data = bytearray(open(os.path.join(work, "pkg.mkv"), "rb").read())
    a, b = video[6]
    data[(a + b) // 2] ^= 0x04
    open(os.path.join(work, "pkg.mkv"), "wb").write(data)
    r = run([linkedbin, "--check", "pkg.mkv"], work, timeout=60)
    assert OK_LINE not in r.stdout and ("Error" in (r.stdout + r.stderr) or r.returncode != 0), r.stdout + r.stderr
    assert "f_000006" in (r.stdout + r.stderr), r.stdout + r.stderr


@pytest.mark.parametrize("damaged", [1, 4])
def test_linked_reference_lists_the_files_the_reference_lists(built, linkedbin, refbin, tmp_path, monkeypatch, damaged):
    """A complaint about a slice stays with the reference's decoder for the rest of the track (parameters::error_message is set once,
    ffv1_frame::Process ends in `return P.Error() ? true : false`, FFV1_Frame.cpp:227): after one damaged frame the unmodified reference
    lists EVERY later frame of the track under "undecodable frame", though their bytes compare equal.  What the user reads from the linked
    binary must be that list, file for file -- and the list of files that differ from their sources, which is the damaged one alone."""
    work = str(tmp_path)
    n = 8
    make_package(work, 96, 64, synth.PIX_RGB16_BE, n, "film")
    monkeypatch.setenv("RCGPU_CHECK_BATCH", "3")
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    import mkv_validator
    blocks = []
    mkv_validator.validate(os.path.join(work, "pkg.mkv"), on_block=lambda trk, t, a, b: blocks.append((trk, a, b)))
    video = [(a, b) for trk, a, b in blocks if trk == 1]
    data = bytearray(open(os.path.join(work, "pkg.mkv"), "rb").read())
    a, b = video[damaged]
    data[(a + b) // 2] ^= 0x04
    open(os.path.join(work, "pkg.mkv"), "wb").write(data)

    def lists(r):
        out, cur = {}, None
        for ln in (r.stdout + r.stderr).replace("\r", "\n").split("\n"):
            if ln.startswith("Error:"):
                cur = ln.strip(); out[cur] = []
            elif cur and ln.startswith("       "):
                out[cur].append(ln.strip())
            elif ln.strip():
                cur = None
        return r.returncode != 0, out
    want = lists(run([refbin, "--check", "pkg.mkv"], work, timeout=120))
    got = lists(run([linkedbin, "--check", "pkg.mkv"], work, timeout=120))
    assert want[0] and got == want, (got, want)
    assert any(len(v) == n - damaged for v in want[1].values()) and any(v == ["pkg/img/f_%06d.dpx" % damaged] for v in want[1].values()), want


@pytest.mark.parametrize("seed", range(int(os.environ.get("RCGPU_SOAK_DAMAGE_FROM", "0")), int(os.environ.get("RCGPU_SOAK_DAMAGE", "1"))))      # soak: RCGPU_SOAK_DAMAGE=40 (50 s a seed)
def test_linked_reference_says_what_the_reference_says_about_damaged_files(built, linkedbin, refbin, tmp_path, monkeypatch, seed):
    """Route C under damage of every kind, held to the unmodified reference: a Matroska file with 10 video frames and audio, then eight
    damaged copies per seed -- a bit flipped inside a frame, in a frame's last bytes (slice sizes, CRCs), in the bytes in front of a frame
    (EBML heads: the look-ahead of oracle/route_c_matroska_cpp.patch walks them), anywhere in the file, a run of zeros inside a frame, two
    damaged frames -- and `--check` by both binaries: the same exit code and the same files under the same complaints.  (A copy the
    unmodified reference itself does not survive -- killed by a signal or out of time -- is not held against anybody.)"""
    work = str(tmp_path)
    n = 10
    make_package(work, 96, 64, [synth.PIX_RGB16_BE, synth.PIX_RGB10_FILLEDA_BE, synth.PIX_RGBA16_LE][seed % 3], n, "film", audio=(2, 16, 48000, 20000))
    monkeypatch.setenv("RCGPU_CHECK_BATCH", ["3", "4", "16"][seed % 3])
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    import mkv_validator
    blocks = []
    mkv_validator.validate(os.path.join(work, "pkg.mkv"), on_block=lambda trk, t, a, b: blocks.append((trk, a, b)))
    video = [(a, b) for trk, a, b in blocks if trk == 1]
    good = open(os.path.join(work, "pkg.mkv"), "rb").read()
    rng = np.random.default_rng(700 + seed)

    def lists(r):
        out, cur = {}, None
        for ln in (r.stdout + r.stderr).replace("\r", "\n").split("\n"):
            if ln.startswith("Error:") or ln.startswith("Warning:"):
                cur = ln.strip(); out.setdefault(cur, [])
            elif cur and ln.startswith("       "):
                out[cur].append(ln.strip())
            elif ln.strip():
                cur = None
        return r.returncode, OK_LINE in r.stdout, out
    compared = 0
    for kind in range(8):
        d = bytearray(good)
        a, b = video[int(rng.integers(0, n))]
        if kind == 0:
            d[int(rng.integers(a, b))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d[b - 1 - int(rng.integers(0, 8))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            d[a - 1 - int(rng.integers(0, 12))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3:
            d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 4:
            at = int(rng.integers(a, b - 16)); d[at:at + 16] = bytes(16)
        elif kind == 5:
            a2, b2 = video[int(rng.integers(0, n))]
            d[(a + b) // 2] ^= 4; d[(a2 + b2) // 3] ^= 16
        elif kind == 6:
            d[a + int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8))          # the key frame bit and the first slice header
        else:
            d[int(rng.integers(video[0][0] - 200, video[0][0]))] ^= 1 << int(rng.integers(0, 8))      # CodecPrivate, track entries, the first cluster's head
        open(os.path.join(work, "pkg.mkv"), "wb").write(d)
        try:
            ref = run([refbin, "--check", "pkg.mkv"], work, timeout=60)
        except subprocess.TimeoutExpired:
            continue
        if ref.returncode < 0:
            continue
        got = run([linkedbin, "--check", "pkg.mkv"], work, timeout=120)
        if lists(got) != lists(ref) and os.environ.get("RCGPU_DAMAGE_DUMP"):            # (a box without a debugger: what both said, and the damaged file)
            open(os.environ["RCGPU_DAMAGE_DUMP"] + "_%d_%d.txt" % (seed, kind), "w").write("LINKED rc %d\n%s\n%s\nREFERENCE rc %d\n%s\n%s\n" % (got.returncode, got.stdout, got.stderr, ref.returncode, ref.stdout, ref.stderr))
            open(os.environ["RCGPU_DAMAGE_DUMP"] + "_%d_%d.diff" % (seed, kind), "w").write(repr([(i, good[i], d[i]) for i in range(len(d)) if d[i] != good[i]]) + " blocks " + repr(video))
        assert lists(got) == lists(ref), (seed, kind, lists(got), lists(ref), got.stderr[-4000:])
        compared += 1
    assert compared >= 5


def test_linked_reference_fails_what_the_reference_fails_to_rebuild(built, linkedbin, refbin, tmp_path, monkeypatch):
    """146 pixels of 12-bit packed RGB in the 6 slice columns the reference itself chooses: its first slice ends on a block of 8 pixels, the
    others do not, and the unmodified reference rebuilds the picture wrongly (Lib/Transform/Transform.cpp:176-177 registers the merging of
    shared blocks only from a first slice that needs it) -- `--check` of a file it agreed to encode says the files are not the sources.  The
    device decoder would rebuild them right; what the user reads from the linked binary must be the reference's verdict all the same (the
    archive will be decoded by whatever RAWcooked there is): the track is left to the pool (RCGPU_TRACE_KEPT says so).  96 pixels wide,
    where every slice ends on a block, both binaries pass."""
    for width, passes in ((146, False), (96, True)):
        work = str(tmp_path / str(width))
        os.makedirs(work)
        make_package(work, width, 45, synth.PIX_RGB12_PACKED_BE, 4, "film")
        r = run([linkedbin, "--no-check-padding", "--hash", "--no-check", "-y", "pkg"], work, timeout=60)      # encoded on the device (route B)
        assert r.returncode == 0, r.stdout + r.stderr
        want = run([refbin, "--check", "pkg.mkv"], work, timeout=120)
        monkeypatch.setenv("RCGPU_TRACE_KEPT", "1")
        got = run([linkedbin, "--check", "pkg.mkv"], work, timeout=120)
        monkeypatch.delenv("RCGPU_TRACE_KEPT")
        assert (want.returncode == 0 and OK_LINE in want.stdout) == passes, want.stdout + want.stderr
        assert got.returncode == want.returncode and (OK_LINE in got.stdout) == passes, got.stdout + got.stderr
        assert ("not same as files from source" in got.stdout + got.stderr) == (not passes)
        assert ("rebuilds wrongly" in got.stderr) == (not passes), got.stderr


def test_linked_reference_decodes_one_batch_ahead(built, linkedbin, refbin, tmp_path, monkeypatch):
    """Route C one batch ahead: once the first frame has told the batch size, the demuxer announces the batch after the current one too and
    ffv1_frame::Process has the decoder start on it (rcgpu_ffv1_decoder_decode_keep_hint_file, by the blocks' places in the file: frames of
    1.8 MB make the reference map its file anew between the hint and the batch's turn, Matroska.cpp:394-408) and adopts it when its turn
    comes.  Same verdicts as with RCGPU_CHECK_AHEAD=0, for a package that checks and for one with a broken frame in a batch that was
    decoded ahead; the trace says that batches were adopted."""
    work = str(tmp_path)
    n = 14
    make_package(work, 640, 480, synth.PIX_RGB16_BE, n, "film")
    monkeypatch.setenv("RCGPU_CHECK_BATCH", "3")
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=120)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
    monkeypatch.setenv("RCGPU_TRACE_KEPT", "1")
    r = run([linkedbin, "--check", "pkg.mkv"], work, timeout=120)
    assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr
    assert r.stderr.count("waited for the batch decoded ahead") >= 2, r.stderr       # batches 3, 4 and 5 of 3 + 3 + 3 + 3 + 2
    monkeypatch.setenv("RCGPU_CHECK_AHEAD", "0")
    r0 = run([linkedbin, "--check", "pkg.mkv"], work, timeout=120)
    assert r0.returncode == 0 and OK_LINE in r0.stdout and "decoded ahead" not in r0.stderr, r0.stdout + r0.stderr
    monkeypatch.delenv("RCGPU_TRACE_KEPT")
    # a flipped bit inside the 9th video block (third batch: decoded ahead): that frame's error, with or without the look-ahead
    import mkv_validator
    blocks = []
    mkv_validator.validate(os.path.join(work, "pkg.mkv"), on_block=lambda trk, t, a, b: blocks.append((trk, a, b)))
    video = [(a, b) for trk, a, b in blocks if trk == 1]
    data = bytearray(open(os.path.join(work, "pkg.mkv"), "rb").read())
    a, b = video[8]
    data[(a + b) // 2] ^= 0x04
    open(os.path.join(work, "pkg.mkv"), "wb").write(data)

    def verdict(r):
        return r.returncode != 0 or "Error" in (r.stdout + r.stderr), sorted(set(ln.strip() for ln in (r.stdout + r.stderr).replace("\r", "\n").split("\n") if "f_0000" in ln))
    bad0 = verdict(run([linkedbin, "--check", "pkg.mkv"], work, timeout=120))
    monkeypatch.delenv("RCGPU_CHECK_AHEAD")
    bad1 = verdict(run([linkedbin, "--check", "pkg.mkv"], work, timeout=120))
    assert bad0[0] and bad1 == bad0 and any("f_000008" in ln for ln in bad1[1]), (bad0, bad1)


def test_linked_reference_judges_whole_batches_on_the_device(built, linkedbin, refbin, tmp_path, monkeypatch):
    """Route C with the payloads staying on the device (oracle/route_c_filewriter_cpp.patch): frame_writer::FrameCall lets every rebuilt
    file wait for the rest of its batch, and the batch is hashed and compared with the files on disk by rcgpu_ffv1_decoder_verify_kept.
    What the user reads must be what the unmodified reference prints -- for a package that checks, and for sources that were changed
    after the encoding: a byte of a payload, a byte of a header, a byte after the payload, a shorter file, a longer file, a file that is
    gone.  RCGPU_CHECK_DEFER=0 (payloads back to the host, the reference's own MD5 and memcmp) must say the same."""
    work = str(tmp_path)
    n = 13
    make_package(work, 96, 64, synth.PIX_RGB16_BE, n, "film", audio=(2, 16, 48000, 26000))
    monkeypatch.setenv("RCGPU_CHECK_BATCH", "5")
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr

    def verdict(r):
        return r.returncode, sorted(set(ln.strip() for ln in (r.stdout + r.stderr).replace("\r", "\n").split("\n")
                                        if "rror" in ln or "ndecodable" in ln or "f_0000" in ln or OK_LINE in ln))

    def files(i):
        return os.path.join(work, "pkg", "img", "f_%06d.dpx" % i)

    def change(i, fn):
        d = bytearray(open(files(i), "rb").read())
        d = fn(d, int.from_bytes(d[4:8], "big"))
        open(files(i), "wb").write(d)

    def flip(at):
        def f(d, off):
            d[at(d, off)] ^= 0x10
            return d
        return f
    change(2, flip(lambda d, off: off + 1234))                # inside the payload
    change(5, flip(lambda d, off: 40))                        # inside the header (the bytes before the payload)
    change(6, flip(lambda d, off: len(d) - 1))                # the last byte
    change(8, lambda d, off: d[:-7])                          # shorter
    change(9, lambda d, off: d + b"\0\0\0")                  # longer
    os.unlink(files(11))                                      # gone
    # `--check x.mkv` alone judges by the hashes in the reversibility data; with `-o` the rebuilt files are compared with the ones there too
    # (CLI/Main.cpp:505, frame_writer::NoOutputCheck)
    assert verdict(run([refbin, "--check", "pkg.mkv"], work))[0] == 0
    want = verdict(run([refbin, "--check", "pkg.mkv", "-o", "."], work))
    assert OK_LINE not in " ".join(want[1]) and all(any("f_%06d" % i in ln for ln in want[1]) for i in (2, 5, 6, 8, 9, 11)), want
    for defer in ("1", "0"):
        monkeypatch.setenv("RCGPU_CHECK_DEFER", defer)
        for batch in ("5", "64"):
            monkeypatch.setenv("RCGPU_CHECK_BATCH", batch)
            got = verdict(run([linkedbin, "--check", "pkg.mkv", "-o", "."], work, timeout=60))
            assert got == want, (defer, batch, got, want)
            r = run([linkedbin, "--check", "pkg.mkv"], work, timeout=60)
            assert r.returncode == 0 and OK_LINE in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("batch", ["3", "256"])
def test_linked_reference_hashes_the_sources_on_the_device(built, linkedbin, refbin, tmp_path, monkeypatch, batch):
    """Route D: the analysis loop (CLI/Main.cpp:280-292) announces the files it is about to open (oracle/route_d_main_cpp.patch) and
    input_base::Hash (Lib/Utils/FileIO/Input_Base.cpp:54-81, oracle/route_d_input_base_cpp.patch) takes their whole-file MD5s from
    rcgpu_md5_host_batch -- many files side by side on the device -- instead of hashing one after the other on one core.  The
    reversibility data the patched binary writes (it holds every file's MD5) must be the unmodified reference's, byte for byte; files of
    different sizes, a batch size that does not divide the sequence, and the reference's own MD5 under RCGPU_HASH=0."""
    work = str(tmp_path)
    make_package(work, 80, 48, synth.PIX_RGB16_BE, 11, "film", audio=(2, 16, 48000, 9000))
    r = run([refbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    want = open(os.path.join(work, "pkg.rawcooked_reversibility_data"), "rb").read()
    os.unlink(os.path.join(work, "pkg.rawcooked_reversibility_data"))
    for env_hash in ("1", "0"):
        monkeypatch.setenv("RCGPU_HASH", env_hash)
        monkeypatch.setenv("RCGPU_HASH_BATCH", batch)
        r = run([linkedbin, "--hash", "--no-check-padding", "-d", "-y", "pkg"], work, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        got = open(os.path.join(work, "pkg.rawcooked_reversibility_data"), "rb").read()
        os.unlink(os.path.join(work, "pkg.rawcooked_reversibility_data"))
        assert got == want, f"RCGPU_HASH={env_hash}: the reversibility data differs"
    # and the whole round trip with device hashes: encode, then check against them
    monkeypatch.setenv("RCGPU_HASH", "1")
    r = run([linkedbin, "--no-check-padding", "--check", "--hash", "-y", "pkg"], work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr


@pytest.mark.parametrize("hash_too", [True, False])
def test_linked_reference_tests_the_padding_bits_on_the_device(built, linkedbin, refbin, tmp_path, monkeypatch, hash_too):
    """Route D, second half: with --check-padding dpx::ParseBuffer walks every payload looking for non-zero padding bits
    (Lib/Uncompressed/DPX/DPX.cpp:501-608).  The files the analysis loop announces are scanned on the device while they are there for their
    MD5 (rcgpu_analysis_host_batch); the loop is skipped for the clean ones (oracle/route_d_dpx_cpp.patch) and runs as ever for the others.
    The reversibility data -- it holds the padding bits of the frames that have any -- must be the unmodified reference's, byte for byte."""
    work = str(tmp_path)
    n = 11
    make_package(work, 80, 48, synth.PIX_RGB10_FILLEDA_BE, n, "film")
    dirty = (3, 7, 8)
    for i in dirty:
        p = os.path.join(work, "pkg", "img", "f_%06d.dpx" % i)
        d = bytearray(open(p, "rb").read())
        off = int.from_bytes(d[4:8], "big")
        for k in range(off + 3 + 4 * 100 * i, len(d), 4 * (50 + i)):
            d[k] |= 1 + (k % 3 == 0)
        open(p, "wb").write(d)
    args = (["--hash"] if hash_too else []) + ["--check-padding", "-d", "-y", "pkg"]
    r = run([refbin] + args, work)
    assert r.returncode == 0, r.stdout + r.stderr
    want = open(os.path.join(work, "pkg.rawcooked_reversibility_data"), "rb").read()
    os.unlink(os.path.join(work, "pkg.rawcooked_reversibility_data"))
    monkeypatch.setenv("RCGPU_TRACE_ROUTE_D", "1")
    for env_hash, batch in (("1", "4"), ("1", "256"), ("0", "256")):
        monkeypatch.setenv("RCGPU_HASH", env_hash)
        monkeypatch.setenv("RCGPU_HASH_BATCH", batch)
        r = run([linkedbin] + args, work, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        got = open(os.path.join(work, "pkg.rawcooked_reversibility_data"), "rb").read()
        os.unlink(os.path.join(work, "pkg.rawcooked_reversibility_data"))
        assert got == want, f"RCGPU_HASH={env_hash} batch {batch}: the reversibility data differs"
        skipped = [ln for ln in r.stderr.replace("\r", "\n").split("\n") if "rcgpu route D: no padding bit set" in ln]      # the progress indicator shares the line
        # the first file is parsed before the loop that announces the others
        assert len(skipped) == (n - 1 - len(dirty) if env_hash == "1" else 0), r.stderr[-600:]
        assert not any("f_%06d" % i in ln for i in dirty for ln in skipped)
    # and the package made from it checks
    monkeypatch.setenv("RCGPU_HASH", "1")
    r = run([linkedbin, "--check-padding", "--check", "-y", "pkg"] + (["--hash"] if hash_too else []), work, timeout=60)
    assert r.returncode == 0 and OK_LINE in r.stdout and "Error" not in (r.stdout + r.stderr), r.stdout + r.stderr
