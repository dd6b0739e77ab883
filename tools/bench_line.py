"""One bench.py JSON line -> the few numbers an A/B needs.  Usage: python tools/bench_line.py <file> [label]"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(sys.argv[2] if len(sys.argv) > 2 else "", d["value"], "frames/s", d["ms_per_step"], "ms/step", {k: round(v, 1) for k, v in d["roofline"]["kernel_ms_per_step"].items()},
      "oracle", c.get("verified_vs_oracle"), "reference", c.get("verified_by_reference"), "err", c.get("device_error_flags"))
