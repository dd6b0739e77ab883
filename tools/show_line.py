"""Pretty-prints a bench.py JSON line (long strings cut).  Usage: python tools/show_line.py <file>"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(k, v, ind=0):
    if isinstance(v, dict):
        print(" " * ind + k + ":")
        for a, b in v.items():
            show(a, b, ind + 2)
    else:
        t = str(v)
        print(" " * ind + k + ": " + (t if len(t) < 150 else t[:150] + "..."))
show("line", d)
