"""Prints the e2e / host_pipeline records of a bench.py JSON line.  Usage: python tools/e2e_line.py <file>"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"])
hp = d.get("host_pipeline") or {}
print("host_pipeline", {k: hp.get(k) for k in ("value", "seconds", "first_packet_seconds", "h2d_span_seconds", "steady_frames_per_second", "fraction_of_device_resident", "batch_frames", "batches", "packets_identical_to_device_resident_run")})
e = d.get("e2e") or {}
print("e2e", {k: e.get(k) for k in ("value", "seconds", "first_block_identical_to_device_resident_run", "all_blocks_verified", "trace", "error", "skipped")})
print("\n".join(e.get("phases", [])))
