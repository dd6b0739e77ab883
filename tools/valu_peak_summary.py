"""Prints tools/valu_peak's JSON lines as a table (one row per instruction form)."""
import json
import sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"op"')]
seen = {}
for r in rows:
    seen.setdefault((r["op"], r["kind"]), {})[r["waves_per_simd"]] = r
for (op, kind), d in seen.items():
    print("%-38s %-11s" % (op, kind), " ".join("w%d:%4.0f" % (w, d[w]["per_simd_per_s"] / 1e6) for w in sorted(d)),
          " M wave-instr/s/SIMD; cycles/instr of one wave %.2f" % d[1]["counter_ticks_per_instr_wave0"])
