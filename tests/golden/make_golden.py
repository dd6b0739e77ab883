#!/usr/bin/env python3
"""Generates tests/golden/{vectors.json,*.bin}: small seeded inputs, the oracle's outputs, and the real reference's verdict.

Run in the build container (needs oracle/_ref/rawcooked, i.e. /root/reference):  python tests/golden/make_golden.py
Each FFV1 vector: payload (file layout bytes) -> packet(s) from oracle/ffv1_oracle.c; an MKV with those packets + the
reference's own reversibility data is then checked by `rawcooked --check` and decoded by `rawcooked -y` (byte compare).
Each FLAC vector: PCM -> frames from oracle/flac_oracle.c, checked the same way.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob          # noqa: E402
from rawcooked_amd import api, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
OK = "Reversibility was checked, no issue detected."

FFV1 = [  # name, w, h, pixfmt, frames, kind, tiff[, layout flags]
    ("dpx_rgb16be_64x48", 64, 48, synth.PIX_RGB16_BE, 2, "film", False),
    ("dpx_rgb10be_50x38", 50, 38, synth.PIX_RGB10_FILLEDA_BE, 2, "film", False),
    ("dpx_rgb10le_61x35", 61, 35, synth.PIX_RGB10_FILLEDA_LE, 1, "noise", False),
    ("dpx_rgb12be_34x17", 34, 17, synth.PIX_RGB12_FILLEDA_BE, 1, "film", False),
    ("dpx_rgb8_48x32", 48, 32, synth.PIX_RGB8, 1, "film", False),
    ("dpx_rgba16be_32x24", 32, 24, synth.PIX_RGBA16_BE, 1, "film", False),
    ("dpx_y16be_40x24", 40, 24, synth.PIX_Y16_BE, 1, "noise", False),
    ("dpx_y8_40x24", 40, 24, synth.PIX_Y8, 1, "film", False),
    ("tiff_rgb16le_40x30", 40, 30, synth.PIX_RGB16_LE, 2, "film", True),
    ("dpx_rgb16be_flat_96x64", 96, 64, synth.PIX_RGB16_BE, 1, "flat", False),
    # bit-packed flavors.  Widths: the reference only merges the words shared by two slices when slice (0,0) itself ends inside
    # a word (Finalize is registered there, Transform.cpp:176-177,724-725,866-867), so 50..56-wide pictures with an aligned first
    # slice fail in the reference itself; these geometries have either every or the first boundary unaligned.
    ("dpx_rgb12packed_56x38", 56, 38, synth.PIX_RGB12_PACKED_BE, 1, "film", False),
    ("dpx_rgb12packed_vflip_96x40", 96, 40, synth.PIX_RGB12_PACKED_BE, 1, "film", False, synth.FLAG_VFLIP),
    ("dpx_rgba10be_96x40", 96, 40, synth.PIX_RGBA10_FILLEDA_BE, 1, "film", False),
    ("dpx_rgba10le_51x38", 51, 38, synth.PIX_RGBA10_FILLEDA_LE, 1, "noise", False),
    ("dpx_rgba12packed_50x38", 50, 38, synth.PIX_RGBA12_PACKED_BE, 1, "film", False),
    ("dpx_rgba12be_50x38", 50, 38, synth.PIX_RGBA12_FILLEDA_BE, 1, "film", False),
    ("dpx_rgba12le_50x38", 50, 38, synth.PIX_RGBA12_FILLEDA_LE, 1, "noise", False),
    ("dpx_y10a_52x38", 52, 38, synth.PIX_Y10_FILLEDA_BE, 1, "film", False),
    ("dpx_y10a_altern_50x38", 50, 38, synth.PIX_Y10_FILLEDA_BE, 1, "film", False, synth.FLAG_ALTERN),
    ("dpx_y10b_100x38", 100, 38, synth.PIX_Y10_FILLEDB_BE, 1, "noise", False),
    ("dpx_y10b_altern_52x38", 52, 38, synth.PIX_Y10_FILLEDB_BE, 1, "film", False, synth.FLAG_ALTERN),
    ("dpx_y12packed_56x38", 56, 38, synth.PIX_Y12_PACKED_BE, 1, "film", False),
    ("dpx_y12packed_vflip_96x40", 96, 40, synth.PIX_Y12_PACKED_BE, 1, "noise", False, synth.FLAG_VFLIP),
    ("exr_rgb16_72x40", 72, 40, synth.PIX_EXR_RGB16, 2, "film", "exr"),
    ("dpx_rgb16be_coder2_72x40", 72, 40, synth.PIX_RGB16_BE, 2, "film", False, 0, 2),      # -coder 2: transition table carried in the record
    ("dpx_rgb10be_coder2_50x38", 50, 38, synth.PIX_RGB10_FILLEDA_BE, 1, "film", False, 0, 2),
    # the other TIFF flavors of TIFF.cpp:157-166
    ("tiff_rgb8_40x30", 40, 30, synth.PIX_RGB8, 1, "film", True),
    ("tiff_rgb16be_40x30", 40, 30, synth.PIX_RGB16_BE, 1, "film", True),
    ("tiff_rgba8_40x30", 40, 30, synth.PIX_RGBA8, 1, "film", True),
    ("tiff_rgba16le_40x30", 40, 30, synth.PIX_RGBA16_LE, 1, "film", True),
    ("tiff_y8_40x30", 40, 30, synth.PIX_Y8, 1, "film", True),
    ("tiff_y16le_40x30", 40, 30, synth.PIX_Y16_LE, 1, "film", True),
    ("tiff_y16be_40x30", 40, 30, synth.PIX_Y16_BE, 1, "noise", True),      # HALF channels taken as uint16 (Output.cpp:120-122)
]
FLAC = [  # name, ch, bits, rate, samples, kind
    ("wav_2ch16_48k", 2, 16, 48000, 10000, "music"),
    ("wav_6ch24_48k", 6, 24, 48000, 6000, "music"),
    ("wav_1ch8_44k", 1, 8, 44100, 5000, "music"),
    ("wav_2ch24_noise", 2, 24, 96000, 9000, "noise"),
    ("wav_2ch16_midside", 2, 16, 48000, 10000, "stereo"),          # channel assignment 10
    ("wav_2ch24_leftside", 2, 24, 48000, 7000, "stereo_left"),     # 8: the side subframe is 25 bits wide
    ("wav_2ch16_sideright", 2, 16, 44100, 6000, "stereo_right"),   # 9
]


# ---- streams FFmpeg's defaults never produce and the reference's decoder accepts (parameters::Parse, FFV1_Parameters.cpp:23-183,206-253;
# slice::SliceHeader, FFV1_Slice.cpp:158-168): what the device `--check` decoder must take too.  Table sets as run lengths of equal levels.
Q9 = [5, 8, 14, 29, 72]          # FFmpeg's 9-level and 5-level maps for > 8 bit, as in oracle/ffv1_oracle.c
Q5 = [11, 53, 64]
Q11_8 = [1, 1, 3, 7, 20, 96]
Q5_8 = [1, 3, 124]
ONE = [128]                      # one level: the input does not count
SET_FFMPEG_3 = [Q9, Q9, Q9, ONE, ONE]
SET_FFMPEG_5 = [Q9, Q9, Q5, Q5, Q5]
SET_ODD_A = [[2, 5, 9, 30, 82], [7, 121], [1, 2, 4, 8, 16, 32, 65], ONE, ONE]           # 5 x 2 x 7 levels, three inputs
SET_ODD_B = [[3, 125], [10, 20, 98], [40, 88], [6, 122], [1, 127]]                       # five inputs, 2-3 levels each
SET_ODD_C = [[1, 1, 1, 1, 1, 1, 1, 121], [4, 4, 120], [128], [128], [16, 112]]           # q[3] has one level: Is5 is false although q[4] is not trivial (FFV1_Slice.cpp:453)
SET_TINY = [[64, 64], ONE, ONE, ONE, ONE]                                                # 2 contexts
SET_8BIT_5 = [Q11_8, Q11_8, Q5_8, Q5_8, Q5_8]


def ext_random_transitions(seed):
    """a transition table nobody ships: each state moves up by a random step, never to 0"""
    import random
    rnd = random.Random(seed)
    one = [0] * 256
    for i in range(1, 256):
        one[i] = min(255, max(1, i + rnd.randint(1, 24) - (rnd.random() < 0.1) * rnd.randint(0, 12)))
    return one


def ext_initial_states(e, sets, seed, lo=30, hi=226):
    import random
    rnd = random.Random(seed)
    out = {}
    for i in sets:
        n = ob.lib().ffv1o_ext_context_count(__import__("ctypes").byref(e), i) * 32
        out[i] = bytes(rnd.randint(lo, hi) for _ in range(n))
    return out


def ext_cases():
    """name, w, h, pixfmt, frames, kind, tiff, num_h, num_v, ec, ext kwargs, initial-state spec (sets, seed, lo, hi) or None"""
    return [
        ("ext_3sets_split_64x48", 64, 48, synth.PIX_RGB16_BE, 2, "film", False, 3, 2, 1, dict(sets=[SET_FFMPEG_3, SET_FFMPEG_5, SET_ODD_B], set_index=(2, 0, 0)), None),
        ("ext_rgba_3groups_48x32", 48, 32, synth.PIX_RGBA16_BE, 1, "film", False, 2, 2, 1, dict(sets=[SET_ODD_A, SET_FFMPEG_3, SET_ODD_B], set_index=(1, 2, 0)), None),
        ("ext_8sets_50x38", 50, 38, synth.PIX_RGB10_FILLEDA_BE, 1, "film", False, 3, 2, 1,
         dict(sets=[SET_TINY, SET_ODD_A, SET_ODD_B, SET_FFMPEG_3, SET_ODD_C, SET_FFMPEG_5, SET_TINY, SET_ODD_A], set_index=(7, 4, 0)), None),
        ("ext_one_set_tiny_40x30", 40, 30, synth.PIX_RGB16_LE, 1, "noise", True, 2, 2, 0, dict(sets=[SET_TINY], set_index=(0, 0, 0)), None),
        ("ext_states_coded_64x48", 64, 48, synth.PIX_RGB16_BE, 2, "film", False, 2, 2, 1, dict(sets=[SET_ODD_A, SET_ODD_B], set_index=(1, 0, 0)), ([0, 1], 7, 30, 226)),
        ("ext_states_coded_one_of_two_40x24", 40, 24, synth.PIX_Y16_BE, 1, "film", False, 2, 1, 1, dict(sets=[SET_ODD_B, SET_ODD_A], set_index=(1, 0, 0)), ([1], 9, 30, 226)),
        ("ext_random_transitions_72x40", 72, 40, synth.PIX_RGB16_BE, 1, "film", False, 3, 2, 1,
         dict(sets=[SET_FFMPEG_3, SET_FFMPEG_5], set_index=(1, 1, 1), one_state=ext_random_transitions(11)), None),
        ("ext_all_at_once_48x32", 48, 32, synth.PIX_RGBA16_BE, 1, "film", False, 2, 2, 1,
         dict(sets=[SET_ODD_C, SET_ODD_A, SET_ODD_B], set_index=(0, 2, 1), one_state=ext_random_transitions(12)), ([0, 2], 13, 40, 200)),
        ("ext_gray_two_sets_40x24", 40, 24, synth.PIX_Y16_BE, 1, "noise", False, 2, 2, 1, dict(sets=[SET_FFMPEG_5, SET_ODD_A], set_index=(1, 0, 0)), None),
        ("ext_v1_inband_custom_48x32", 48, 32, synth.PIX_RGB16_BE, 2, "film", False, 1, 1, 0, dict(version=1, sets=[SET_ODD_B], one_state=ext_random_transitions(14)), None),
        ("ext_v0_rgb8_48x32", 48, 32, synth.PIX_RGB8, 1, "film", False, 1, 1, 0, dict(version=0, sets=[SET_8BIT_5]), None),
        # valid streams the device decoder does not take: the reference decodes them, route C must fall back to the slice pool.  The set index is
        # a per-SLICE field (the device wants one tuple per stream and finds out while decoding); intra = 0 with key frames only (refused up front)
        ("ext_unsup_per_slice_sets_64x48", 64, 48, synth.PIX_RGB16_BE, 2, "film", False, 3, 2, 1, dict(sets=[SET_ODD_A, SET_ODD_B], set_index=(1, 0, 0), set_index_alt=(0, 1, 0)), None),
        ("ext_unsup_intra0_64x48", 64, 48, synth.PIX_RGB16_BE, 2, "film", False, 2, 2, 1, dict(sets=[SET_FFMPEG_3, SET_FFMPEG_5], set_index=(1, 1, 1), intra=0), None),
    ]


def build_ext(case):
    name, w, h, pixfmt, nframes, kind, tiff, nh, nv, ec, kw, init = case
    e = ob.stream_ext(**kw)
    if init:
        sets, seed, lo, hi = init
        e = ob.stream_ext(initial_states=ext_initial_states(e, sets, seed, lo, hi), **kw)
    return e


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=120)


def main():
    vectors = {"ffv1": [], "flac": []}
    for name, w, h, pixfmt, nframes, kind, tiff, *rest in FFV1:
        flags = rest[0] if rest else 0
        coder = rest[1] if len(rest) > 1 else 1
        work = tempfile.mkdtemp()
        os.makedirs(work + "/seq")
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        files = []
        for i in range(nframes):
            comp = synth.components(w, h, nc, bits, kind, seed=31 * i + 5)
            data = synth.exr_file(comp, trailer=b"xyz") if tiff == "exr" else synth.tiff_file(comp, pixfmt, trailer=b"xyz") if tiff else synth.dpx_file(comp, pixfmt, frame_index=i, flags=flags)
            fn = work + "/seq/f_%06d.%s" % (i, "exr" if tiff == "exr" else "tif" if tiff else "dpx")
            open(fn, "wb").write(data)
            files.append(fn)
        ri = run([REF, "--info", "--no-encode", "--no-check-padding", "--check", "-y", "seq"], work)      # prints the flavor string of every track
        r = run([REF] + (["-coder", str(coder)] if coder != 1 else []) + ["--hash", "--no-check-padding", "--check", "-d", "-y", "seq"], work)      # --check: the reference refuses EXR without it (Main.cpp:121-127)
        assert r.returncode == 0, r.stderr
        first = open(files[0], "rb").read()
        info = api.exr_probe(first) if tiff == "exr" else api.tiff_probe(first) if tiff else api.dpx_probe(first)
        slices = int(r.stdout.split("-slices ")[1].split()[0])
        assert slices == info.slices and info.flags == flags and ("-vf vflip" in r.stdout) == bool(flags & synth.FLAG_VFLIP)
        assert (" " + info.flavor.decode() + "\n") in ri.stdout + ri.stderr, (info.flavor, ri.stdout, ri.stderr)
        nh, nv = api.slices_to_grid(slices)
        assert ("-coder %d " % coder) in r.stdout
        p = ob.Params(w, h, pixfmt, nh, nv, 1, 1, flags, coder)
        rec = ob.config_record(p)
        mux = api.MkvMuxer(work + "/seq.mkv")
        t = mux.add_video(rec, w, h, 24, 1)
        mux.add_attachment("RAWcooked reversibility data", open(work + "/seq.rawcooked_reversibility_data", "rb").read())
        mux.begin()
        entry = {"name": name, "width": w, "height": h, "pixfmt": pixfmt, "num_h": nh, "num_v": nv, "line_bytes": info.line_bytes,
                 "flags": flags, "coder": coder, "flavor": info.flavor.decode(), "slices": slices, "config_record": rec.hex(), "frames": []}
        for i, fn in enumerate(files):
            b = open(fn, "rb").read()
            payload = b[info.data_offset:info.data_offset + info.data_size]
            pkt = ob.encode_payload(p, payload, info.line_bytes)
            mux.write_block(t, i * 1000000000 // 24, pkt)
            open(os.path.join(HERE, f"ffv1_{name}_{i}.payload.bin"), "wb").write(payload)
            open(os.path.join(HERE, f"ffv1_{name}_{i}.packet.bin"), "wb").write(pkt)
            entry["frames"].append({"payload": f"ffv1_{name}_{i}.payload.bin", "packet": f"ffv1_{name}_{i}.packet.bin",
                                    "payload_sha256": hashlib.sha256(payload).hexdigest(), "packet_sha256": hashlib.sha256(pkt).hexdigest()})
        mux.close()
        r = run([REF, "--check", "seq.mkv"], work)
        ok = r.returncode == 0 and OK in r.stdout
        r2 = run([REF, "-y", "seq.mkv"], work)
        same = all(open(f, "rb").read() == open(work + "/seq.mkv.RAWcooked/seq/" + os.path.basename(f), "rb").read() for f in files)
        entry["reference_check"] = bool(ok and r2.returncode == 0 and same)
        assert entry["reference_check"], (name, r.stdout, r.stderr)
        vectors["ffv1"].append(entry)
        shutil.rmtree(work)
        print("ffv1", name, "reference ok")
    vectors["ffv1_ext"] = []
    for case in ext_cases():
        name, w, h, pixfmt, nframes, kind, tiff, nh, nv, ec, kw, init = case
        work = tempfile.mkdtemp()
        os.makedirs(work + "/seq")
        bits, nc, _, _ = synth.PIX_INFO[pixfmt]
        files = []
        for i in range(nframes):
            comp = synth.components(w, h, nc, bits, kind, seed=17 * i + 3)
            data = synth.tiff_file(comp, pixfmt, trailer=b"xyz") if tiff else synth.dpx_file(comp, pixfmt, frame_index=i)
            fn = work + "/seq/f_%06d.%s" % (i, "tif" if tiff else "dpx")
            open(fn, "wb").write(data)
            files.append(fn)
        r = run([REF, "--hash", "--no-check-padding", "--check", "-d", "-y", "seq"], work)
        assert r.returncode == 0, r.stderr
        first = open(files[0], "rb").read()
        info = api.tiff_probe(first) if tiff else api.dpx_probe(first)
        e = build_ext(case)
        # the description, in a form tests/ rebuild it from without this script (ob.stream_ext(**ext)); initial states as files
        ext_json = dict(kw)
        if init:
            ext_json["initial_states"] = {}
            for i, buf in zip(init[0], e._keep):
                open(os.path.join(HERE, f"ffv1_{name}.init{i}.bin"), "wb").write(buf.raw)
                ext_json["initial_states"][str(i)] = f"ffv1_{name}.init{i}.bin"
        p = ob.with_ext(ob.Params(w, h, pixfmt, nh, nv, ec), e)
        rec = ob.config_record(p)
        assert bool(rec) == (e.version == 3)
        mux = api.MkvMuxer(work + "/seq.mkv")
        t = mux.add_video(rec, w, h, 24, 1)
        mux.add_attachment("RAWcooked reversibility data", open(work + "/seq.rawcooked_reversibility_data", "rb").read())
        mux.begin()
        entry = {"name": name, "width": w, "height": h, "pixfmt": pixfmt, "num_h": nh, "num_v": nv, "line_bytes": info.line_bytes, "ec": ec, "flags": 0,
                 "version": e.version, "flavor": info.flavor.decode(), "config_record": "", "ext": ext_json, "frames": []}
        if len(rec) <= 4096:
            entry["config_record"] = rec.hex()
        else:                                                    # coded initial states: 32 symbols per context
            open(os.path.join(HERE, f"ffv1_{name}.record.bin"), "wb").write(rec)
            entry["config_record_file"] = f"ffv1_{name}.record.bin"
        for i, fn in enumerate(files):
            b = open(fn, "rb").read()
            payload = b[info.data_offset:info.data_offset + info.data_size]
            pkt = ob.encode_payload(p, payload, info.line_bytes)
            code, back = ob.decode_stream(ob.Params(w, h, pixfmt), rec, pkt, info.line_bytes)      # the oracle's own reading of what it wrote
            assert code == 0 and back == payload, (name, code)
            mux.write_block(t, i * 1000000000 // 24, pkt)
            open(os.path.join(HERE, f"ffv1_{name}_{i}.payload.bin"), "wb").write(payload)
            open(os.path.join(HERE, f"ffv1_{name}_{i}.packet.bin"), "wb").write(pkt)
            entry["frames"].append({"payload": f"ffv1_{name}_{i}.payload.bin", "packet": f"ffv1_{name}_{i}.packet.bin",
                                    "payload_sha256": hashlib.sha256(payload).hexdigest(), "packet_sha256": hashlib.sha256(pkt).hexdigest()})
        mux.close()
        r = run([REF, "--check", "seq.mkv"], work)
        ok = r.returncode == 0 and OK in r.stdout
        r2 = run([REF, "-y", "seq.mkv"], work)
        same = all(open(f, "rb").read() == open(work + "/seq.mkv.RAWcooked/seq/" + os.path.basename(f), "rb").read() for f in files)
        entry["reference_check"] = bool(ok and r2.returncode == 0 and same)
        assert entry["reference_check"], (name, r.stdout, r.stderr)
        vectors["ffv1_ext"].append(entry)
        shutil.rmtree(work)
        print("ffv1_ext", name, "reference ok", len(rec), "byte record")
    for name, ch, bits, rate, n, kind in FLAC:
        work = tempfile.mkdtemp()
        os.makedirs(work + "/aud")
        wav = synth.wav_file(synth.pcm_samples(n, ch, bits, rate, kind), bits, rate)
        open(work + "/aud/a.wav", "wb").write(wav)
        info = api.wav_probe(wav)
        pcm = wav[info.data_offset:info.data_offset + info.data_size]
        frames, cp = ob.flac_encode(ch, rate, bits, pcm, 0, 8)
        r = run([REF, "--hash", "-d", "-y", "aud"], work)
        assert r.returncode == 0, r.stderr
        mux = api.MkvMuxer(work + "/aud.mkv")
        t = mux.add_audio(cp, ch, rate, bits)
        mux.add_attachment("RAWcooked reversibility data", open(work + "/aud.rawcooked_reversibility_data", "rb").read())
        mux.begin()
        B = int.from_bytes(cp[8:10], "big")
        for i, f in enumerate(frames):
            mux.write_block(t, i * B * 1000000000 // rate, f)
        mux.close()
        r = run([REF, "--check", "aud.mkv"], work)
        ok = r.returncode == 0 and OK in r.stdout
        assert ok, (name, r.stdout, r.stderr)
        open(os.path.join(HERE, f"flac_{name}.pcm.bin"), "wb").write(pcm)
        open(os.path.join(HERE, f"flac_{name}.frames.bin"), "wb").write(b"".join(frames))
        vectors["flac"].append({"name": name, "channels": ch, "bits": bits, "rate": rate, "pcm": f"flac_{name}.pcm.bin",
                                "frames": f"flac_{name}.frames.bin", "frame_sizes": [len(f) for f in frames], "codec_private": cp.hex(),
                                "frames_sha256": hashlib.sha256(b"".join(frames)).hexdigest(), "reference_check": True})
        shutil.rmtree(work)
        print("flac", name, "reference ok")
    json.dump(vectors, open(os.path.join(HERE, "vectors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
