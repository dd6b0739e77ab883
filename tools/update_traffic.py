#!/usr/bin/env python3
"""profiles/traffic.json from the round's PMC summaries, tied to the code they were measured on.

  python tools/update_traffic.py <round tag, e.g. r05> [--enc-batch 336] [--check-batch 1600]

Reads profiles/<tag>_pmc_fetch_size.csv + _pmc_write_size.csv (encode half, `bash tools/round.sh encprof`), profiles/<tag>_check_pmc_*.csv
(check half, `bash tools/round.sh checkprof`) and profiles/<tag>_sq_counters.csv (SQ_INSTS_VALU at batch 336), computes HBM bytes per 4K
frame for every kernel, and records the sha256 of rawcooked_amd/csrc/ffv1_gpu.hip and ffv1_check.hip AS THEY ARE NOW: run it right after the
passes, on the tree they ran on.  bench.py prints `traffic: null, traffic_stale: true` for a kernel whose source has changed since, and
tests/test_host.py::test_traffic_json_describes_this_code fails until the passes are taken again.

Counter semantics (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KB (x 1024); gfx950 reports HALF
the bytes of wide coalesced reads (calibrated in round 1 on k_unpack: 53.08 MB of input per frame read as 26.5), so the stream-reading kernels'
FETCH_SIZE is doubled (`fetch_x2`), while kernels whose reads are 32-byte gathers and 4-byte symbol reads are taken as reported."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
SOURCES = ["rawcooked_amd/csrc/ffv1_gpu.hip", "rawcooked_amd/csrc/ffv1_check.hip"]
FETCH_X2 = {"k_rangecode", "k_rc_range", "k_rangecode<true>", "k_model", "k_gather", "k_footer"}      # wide coalesced stream readers


def table(path, col):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        name = r["kernel"].replace("<false>", "").replace("k_dec_slices<true>", "k_dec_slices")
        out[name] = (int(r["launches"]), float(r[col]) * 1024.0)       # total bytes over all launches
    return out


def main():
    tag = sys.argv[1]
    enc_batch = int(sys.argv[sys.argv.index("--enc-batch") + 1]) if "--enc-batch" in sys.argv else 336
    chk_batch = int(sys.argv[sys.argv.index("--check-batch") + 1]) if "--check-batch" in sys.argv else 1600
    tj = os.path.join(P, "traffic.json")
    d = json.load(open(tj))
    for half, batch, src in (("", enc_batch, "ffv1_gpu.hip"), ("check_", chk_batch, "ffv1_check.hip")):
        f = table(os.path.join(P, f"{tag}_{half}pmc_fetch_size.csv"), "FETCH_SIZE_sum_KB")
        w = table(os.path.join(P, f"{tag}_{half}pmc_write_size.csv"), "WRITE_SIZE_sum_KB")
        for k in sorted(set(f) & set(w)):
            if half and k != "k_dec_slices" or not half and k.startswith("k_dec"):
                continue
            if k in ("k_copy8", "k_scan", "k_dec_split"):
                continue
            launches = f[k][0]
            per_step = 1 if half else max(1, round(launches / max(1, f.get("k_model", (1, 0))[0])))      # launches of the kernel per step of `batch` frames
            steps = launches / per_step
            fetch = f[k][1] * (2 if k in FETCH_X2 else 1)
            per_frame = (fetch + w[k][1]) / steps / batch
            d.setdefault(k, {})
            d[k].update({"per_frame_bytes": int(per_frame), "fetch_per_frame_bytes": int(fetch / steps / batch), "write_per_frame_bytes": int(w[k][1] / steps / batch),
                         "measured_at_batch": batch, "launches_per_step": per_step, "fetch_x2": k in FETCH_X2, "source": src,
                         "method": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/{tag}_{half}pmc_*.csv, KB as reported x 1024"
                                   + ("; FETCH_SIZE x 2: gfx950 reports half the bytes of wide coalesced reads" if k in FETCH_X2 else "; FETCH_SIZE as reported: 32-byte gathers and 4-byte symbol reads") + ")"})
    sq = os.path.join(P, f"{tag}_sq_counters.csv")
    if os.path.exists(sq):
        v = {}
        for r in csv.DictReader(open(sq)):
            if r["counter"] == "SQ_INSTS_VALU":
                v[r["kernel"].replace("<false>", "")] = float(r["sum"]) / max(1, int(r["launches"]))      # per launch, summed over the chip
        steps_model = 1
        if "k_resolve" in v and "k_rangecode" in v:
            nl = d.get("k_resolve", {}).get("launches_per_step", 32)
            whole = (v["k_resolve"] + v["k_rangecode"]) * nl / enc_batch + v.get("k_model", 0) / enc_batch
            d.setdefault("valu", {})
            d["valu"]["wave_instr_per_frame_whole"] = int(whole)
            d["valu"]["note"] = (f"SQ_INSTS_VALU per 4K frame (profiles/{tag}_sq_counters.csv, batch {enc_batch}, run-on build): k_resolve {v['k_resolve'] * nl / enc_batch / 1e6:.1f} M, "
                                 f"k_rangecode {v['k_rangecode'] * nl / enc_batch / 1e6:.1f} M (whole-slice coder), k_model {v.get('k_model', 0) / enc_batch / 1e6:.1f} M; "
                                 "wave_instr_per_frame_split and the peaks are round 3's (profiles/r03_sq_counters.csv, profiles/r03_valu_peak.txt): a SIMD issues 1.05 G wave64 instructions/s of "
                                 "add/sub/logic/right-shift/mov but 0.58 G/s of everything else, a single wavefront never more than one per 4.7 cycles (8.1 when dependent)")
    # the dominant kernel at the other encode configurations of the bench, from passes of their own (tools/profile_enc.sh <tag>_576 --slices 576 --batch 40,
    # <tag>_8k --width 8192 --height 4320 --slices 576 --batch 80): bytes per frame OF THAT SIZE
    d["configs"] = {}
    for cfgtag, key, batch in (("576", "4096x2160/576", 40), ("8k", "8192x4320/576", 80)):
        f = table(os.path.join(P, f"{tag}_{cfgtag}_pmc_fetch_size.csv"), "FETCH_SIZE_sum_KB")
        w = table(os.path.join(P, f"{tag}_{cfgtag}_pmc_write_size.csv"), "WRITE_SIZE_sum_KB")
        if "k_resolve" in f and "k_resolve" in w and "k_model" in f:
            steps = f["k_model"][0]
            d["configs"][key] = {"k_resolve": {"per_frame_bytes": int((f["k_resolve"][1] + w["k_resolve"][1]) / steps / batch), "measured_at_batch": batch,
                                                "launches_per_step": round(f["k_resolve"][0] / steps),
                                                "method": f"profiles/{tag}_{cfgtag}_pmc_*.csv, as the 4K / 64-slice passes"}}
    d["sources"] = {s: hashlib.sha256(open(os.path.join(ROOT, s), "rb").read()).hexdigest() for s in SOURCES}
    sys.path.insert(0, ROOT)
    from rawcooked_amd import devcode
    d["device_code_sha256"] = devcode.fatbin_sha256(os.path.join(ROOT, "rawcooked_amd", "librcgpu.so"))      # the kernels themselves, as built from those sources
    d["measured_on"] = f"{tag}: the tree whose kernel sources have the sha256 under `sources` (built: device code `device_code_sha256`)"
    json.dump(d, open(tj, "w"), indent=1)
    for k in ("k_resolve", "k_rangecode", "k_model", "k_dec_slices"):
        if k in d:
            print(k, d[k].get("per_frame_bytes"), "bytes per 4K frame")
    print("sources:", d["sources"])


if __name__ == "__main__":
    main()
