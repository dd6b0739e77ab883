"""Route C and the streams the device does not take: the reference with librcgpu.so linked in (oracle/route_c_ffv1_frame_cpp.patch,
ffv1_frame::Process, Lib/CoDec/FFV1/FFV1_Frame.cpp:134-228) must give the verdict of the unmodified reference on EVERY file -- with the
device decoder where it takes the stream, on the reference's own slice pool where it does not (RCGPU_FFV1_UNSUPPORTED, no device, no
CodecPrivate), never with an error of its own.

The packages are the golden `ffv1_ext` streams (what parameters::Parse accepts and FFmpeg's defaults never produce; each one checked by the real
reference when it was generated, tests/golden/make_golden.py) muxed with the reference's own reversibility data.  The first test runs
where there is no device (the build container: every track falls back); the second on the GPU box."""
import json
import os
import subprocess

import pytest

import oracle_binding as ob
from rawcooked_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))
OK_LINE = "Reversibility was checked, no issue detected."      # test2.sh:37,69


def run(cmd, cwd, env=None, timeout=60):
    e = dict(os.environ)
    e.update(env or {})
    for attempt in range(2):                                   # one retry: the reference's thread pool can lose its shutdown wake-up (tests/test_gpu_e2e.py::run)
        try:
            return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL, env=e)
        except subprocess.TimeoutExpired:
            if attempt:
                raise


def record_of(v):
    return open(os.path.join(G, v["config_record_file"]), "rb").read() if "config_record_file" in v else bytes.fromhex(v.get("config_record", ""))


def package(v, work, refbin):
    """the vector's source files (regenerated from their seeds and compared with the committed payloads), the reference's reversibility data
    for them, and an MKV that carries the committed packets"""
    os.makedirs(os.path.join(work, "seq"))
    bits, nc, _, _ = synth.PIX_INFO[v["pixfmt"]]
    tiff = v["flavor"].startswith("TIFF")
    kind = "noise" if "noise" in v["name"] or v["name"] in ("ext_one_set_tiny_40x30", "ext_gray_two_sets_40x24") else "film"
    files = []
    for i, fr in enumerate(v["frames"]):
        comp = synth.components(v["width"], v["height"], nc, bits, kind, seed=17 * i + 3)
        data = synth.tiff_file(comp, v["pixfmt"], trailer=b"xyz") if tiff else synth.dpx_file(comp, v["pixfmt"], frame_index=i)
        info = api.tiff_probe(data) if tiff else api.dpx_probe(data)
        assert data[info.data_offset:info.data_offset + info.data_size] == open(os.path.join(G, fr["payload"]), "rb").read(), "the seeds of make_golden.py"
        fn = os.path.join(work, "seq", "f_%06d.%s" % (i, "tif" if tiff else "dpx"))
        open(fn, "wb").write(data)
        files.append(fn)
    r = run([refbin, "--hash", "--no-check-padding", "--check", "-d", "-y", "seq"], work)
    assert r.returncode == 0, r.stdout + r.stderr
    mux = api.MkvMuxer(os.path.join(work, "seq.mkv"))
    t = mux.add_video(record_of(v), v["width"], v["height"], 24, 1)
    mux.add_attachment("RAWcooked reversibility data", open(os.path.join(work, "seq.rawcooked_reversibility_data"), "rb").read())
    mux.begin()
    for i, fr in enumerate(v["frames"]):
        mux.write_block(t, i * 1000000000 // 24, open(os.path.join(G, fr["packet"]), "rb").read())
    mux.close()
    return files


def check_both_ways(v, work, refbin, linkedbin):
    want = run([refbin, "--check", "seq.mkv"], work)
    assert want.returncode == 0 and OK_LINE in want.stdout, want.stdout + want.stderr
    got = run([linkedbin, "--check", "seq.mkv"], work, env={"RCGPU_TRACE_KEPT": "1"})
    assert got.returncode == 0 and OK_LINE in got.stdout and "Error" not in got.stdout + got.stderr, got.stdout + got.stderr
    off = run([linkedbin, "--check", "seq.mkv"], work, env={"RCGPU_CHECK": "0"})
    assert (off.returncode, OK_LINE in off.stdout) == (0, True)
    return got.stderr


@pytest.mark.parametrize("v", VEC["ffv1_ext"], ids=lambda v: v["name"])
def test_without_a_device_every_track_stays_on_the_slice_pool(built, refbin, linkedbin, tmp_path, v):
    if api.lib().rcgpu_device_count() > 0:
        pytest.skip("a device is present: the GPU form of this test runs instead")
    package(v, str(tmp_path), refbin)
    trace = check_both_ways(v, str(tmp_path), refbin, linkedbin)
    if v["version"] == 3:                                     # (version 0 / 1: no CodecPrivate -- the hook is not reached, FFV1_Frame.cpp:159)
        assert "track left to the slice pool" in trace and ("no HIP device" in trace or "not decoded on the device" in trace), trace
    # a flipped bit in a frame is the reference's own error, in the reference's own words
    data = bytearray(open(tmp_path / "seq.mkv", "rb").read())
    data[len(data) - len(open(os.path.join(G, v["frames"][-1]["packet"]), "rb").read()) // 2] ^= 0x04
    open(tmp_path / "seq.mkv", "wb").write(data)
    want = run([refbin, "--check", "seq.mkv"], str(tmp_path))
    got = run([linkedbin, "--check", "seq.mkv"], str(tmp_path))
    assert OK_LINE not in want.stdout and OK_LINE not in got.stdout
    assert sorted(ln for ln in (want.stdout + want.stderr).split("\n") if "rror" in ln) == sorted(ln for ln in (got.stdout + got.stderr).split("\n") if "rror" in ln)


@pytest.mark.gpu
@pytest.mark.parametrize("v", VEC["ffv1_ext"], ids=lambda v: v["name"])
def test_the_device_takes_what_it_can_and_leaves_the_rest_to_the_slice_pool(built, refbin, linkedbin, tmp_path, v):
    files = package(v, str(tmp_path), refbin)
    trace = check_both_ways(v, str(tmp_path), refbin, linkedbin)
    if "unsup_intra0" in v["name"]:
        assert "track left to the slice pool" in trace and "intra = 0" in trace, trace
    elif "unsup_per_slice" in v["name"]:                      # found while decoding: two frames are handed over one by one, then the track
        assert trace.count("frame left to the slice pool") == 2 and "track left" not in trace, trace
    else:
        assert "left to the slice pool" not in trace, trace
    # the full decode through the same route: byte-identical files
    r = run([linkedbin, "-y", "seq.mkv"], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    for fn in files:
        assert open(fn, "rb").read() == open(os.path.join(str(tmp_path), "seq.mkv.RAWcooked", "seq", os.path.basename(fn)), "rb").read(), fn
    # a flipped bit in the last frame: the verdict and its wording are the unmodified reference's
    data = bytearray(open(tmp_path / "seq.mkv", "rb").read())
    data[len(data) - len(open(os.path.join(G, v["frames"][-1]["packet"]), "rb").read()) // 2] ^= 0x04
    open(tmp_path / "seq.mkv", "wb").write(data)
    want = run([refbin, "--check", "seq.mkv"], str(tmp_path))
    got = run([linkedbin, "--check", "seq.mkv"], str(tmp_path))
    assert OK_LINE not in want.stdout and OK_LINE not in got.stdout, got.stdout + got.stderr
    assert sorted(ln for ln in (want.stdout + want.stderr).split("\n") if "rror" in ln) == sorted(ln for ln in (got.stdout + got.stderr).split("\n") if "rror" in ln)
