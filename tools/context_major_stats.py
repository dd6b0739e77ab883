"""What a context-major pass of k_resolve would have to move: measured on the oracle's symbols of the bench's
content, not estimated.  One 512x270 slice of a 4096x2160 RGB16 "film" frame (and "noise"), cut into N segments in coding order as the
encoder cuts it; per segment: distinct (plane set, context) keys = state records a context-major pass gathers and writes back once each,
the longest run of one context (the serial part of a lane that walks a run), and what has to be staged for the pass to write its
decisions in coding order (the segment's decision stream).  Today's k_resolve moves 0.967 records per sample each way (64-symbol chunks).
    python tools/context_major_stats.py > profiles/r06_context_major.jsonl        (CPU only: oracle/liboracle.so)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    import oracle_binding as ob
    from rawcooked_amd import synth
    W, H = 4096, 2160
    for kind in ("film", "noise"):
        fr = bench.make_frames(torch, 1, W, H, kind, 0, torch.device("cpu"))[0].numpy().tobytes()
        p = ob.Params(W, H, synth.PIX_RGB16_BE, 8, 8, 1, 1)
        planes = ob.unpack(p, fr, W * 6, 3)
        w, h = 512, 270
        sym, dec, raw = ob.trace_slice(p, planes, 3, 3, w * h * 3)
        key = (sym >> 30).astype(np.int64) * 8192 + ((sym >> 17) & 0x1FFF)
        d = (sym & 0x1FFFF).astype(np.int64)
        d = np.where(d >= 0x10000, d - 0x20000, d)
        a = np.abs(d)
        e = np.zeros(len(a), dtype=np.int64)
        nz = a > 0
        e[nz] = np.floor(np.log2(a[nz])).astype(np.int64)
        ndec = np.where(nz, 2 * e + 3, 1)
        n = len(sym)
        chunks = key[:n // 64 * 64].reshape(-1, 64)
        today = float(np.mean([len(np.unique(r)) for r in chunks])) / 64
        for nseg in (32, 64, 128, 256):
            q = ((n + nseg - 1) // nseg + 63) & ~63
            rec, longest, stream_kb, classes = [], [], [], []
            for j in range(nseg):
                k = key[j * q:(j + 1) * q]
                if not len(k):
                    continue
                u, c = np.unique(k, return_counts=True)
                rec.append(len(u) / len(k)); longest.append(int(c.max())); classes.append(int(c.max()))
                stream_kb.append(int(ndec[j * q:(j + 1) * q].sum()) * 9 / 8 / 1024)
            print(json.dumps({"content": kind, "slice": "512x270 of 4096x2160 RGB16, 64 slices", "segments": nseg, "symbols_per_segment": q,
                              "records_per_sample_each_way_today": round(today, 3), "records_per_sample_each_way_context_major": round(float(np.mean(rec)), 3),
                              "longest_run_of_one_context_mean": round(float(np.mean(longest)), 1), "longest_run_max": int(max(longest)),
                              "decision_stream_of_a_segment_kb_mean": round(float(np.mean(stream_kb)), 1), "symbols_of_a_segment_kb": round(q * 4 / 1024, 1),
                              "histogram_of_10126_keys_kb": 39.6, "fits_160_kb_of_lds": bool(np.mean(stream_kb) + q * 4 / 1024 + 39.6 < 150),
                              "decisions_per_sample": round(float(ndec.mean()), 2)}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
