"""Samples the GPU's shader clock and power (hwmon sysfs) every 50 ms while a command runs; prints min / mean / max and a coarse timeline.
Usage on the GPU box: python tools/power_poll.py -- python bench.py --steps 3 --legs "" --no-verify"""
import glob
import subprocess
import sys
import threading
import time

def find():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("freq1_input", "sclk_hz"), ("power1_average", "power_uw"), ("power1_input", "power_uw"), ("temp1_input", "temp_mc"), ("power1_cap", "cap_uw")):
            p = h + "/" + name
            try:
                open(p).read()
                out.setdefault(key, p)
            except Exception:
                pass
    return out

def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    paths = find()
    print("power_poll: sources", paths, file=sys.stderr)
    samples = []
    stop = threading.Event()
    def poll():
        t0 = time.time()
        while not stop.is_set():
            row = {"t": time.time() - t0}
            for k, p in paths.items():
                try:
                    row[k] = int(open(p).read().strip())
                except Exception:
                    pass
            samples.append(row)
            time.sleep(0.05)
    th = threading.Thread(target=poll); th.start()
    r = subprocess.run(cmd)
    stop.set(); th.join()
    for k in ("sclk_hz", "power_uw", "temp_mc", "cap_uw"):
        v = [s[k] for s in samples if k in s]
        if v:
            print("power_poll: %-9s min %.4g mean %.4g max %.4g (%d samples)" % (k, min(v), sum(v) / len(v), max(v), len(v)), file=sys.stderr)
    step = max(1, len(samples) // 60)
    print("power_poll: timeline (t s, sclk MHz, power W):", " ".join("%.1f:%d:%d" % (s["t"], s.get("sclk_hz", 0) / 1e6, s.get("power_uw", 0) / 1e6) for s in samples[::step]), file=sys.stderr)
    sys.exit(r.returncode)

main()
