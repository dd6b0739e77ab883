"""Hunts the time-outs the e2e tests used to retry (tests/test_gpu_e2e.py::run): runs the reference's steps of a small package in a loop,
each under a time limit, and when one runs into it dumps what every thread of the process was doing -- kernel side from /proc (wchan,
syscall), user side through tools/bin/hang_probe.so (LD_PRELOAD: every thread writes its call stack on SIGUSR2; this image has no
debugger) -- with the addresses resolved against `nm` of the binary.  Usage on the GPU box:
    python tools/hang/hang_hunt.py <rounds> [limit seconds] > gpurun_out/<...>/hang_hunt.txt
"""
import bisect
import ctypes
import os
import shlex
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rawcooked_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
LINKED = os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")
SHIM = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
PROBE = os.path.join(ROOT, "tools", "bin", "hang_probe.so")
libc = ctypes.CDLL(None, use_errno=True)
SYS_tgkill = 234


def symbols(binary):
    out = subprocess.run(["nm", "-C", "--defined-only", binary], capture_output=True, text=True).stdout
    tab = []
    for ln in out.splitlines():
        p = ln.split(" ", 2)
        if len(p) == 3 and p[1] in "TtWw":
            tab.append((int(p[0], 16), p[2]))
    tab.sort()
    return tab


def resolve(tab, off):
    i = bisect.bisect_right(tab, (off, "\xff")) - 1
    return "%s +0x%x" % (tab[i][1], off - tab[i][0]) if i >= 0 else hex(off)


def dump(proc, binary, dump_path, tabs):
    pid = proc.pid
    print("  ---- threads of pid %d (%s)" % (pid, os.path.basename(binary)))
    tids = sorted(int(t) for t in os.listdir("/proc/%d/task" % pid))
    for tid in tids:
        def rd(name):
            try:
                return open("/proc/%d/task/%d/%s" % (pid, tid, name)).read().strip()
            except OSError:
                return "?"
        st = rd("stat").rsplit(")", 1)[-1].split()
        print("  tid %d state %s wchan %s syscall %s" % (tid, st[0] if st else "?", rd("wchan"), " ".join(rd("syscall").split()[:2])))
    open(dump_path, "w").close()
    for tid in tids:
        libc.syscall(SYS_tgkill, pid, tid, signal.SIGUSR2)
        time.sleep(0.02)
    time.sleep(0.5)
    base = {}
    for ln in open("/proc/%d/maps" % pid):
        f = ln.split()
        if len(f) >= 6 and f[5] not in base:
            base[f[5]] = int(f[0].split("-")[0], 16)
    for ln in open(dump_path):
        ln = ln.rstrip()
        if ln.startswith("  #") and "[" in ln:
            addr = int(ln.split()[1], 16)
            mod = ln.split("[", 1)[1].split(" base")[0]
            real = os.path.realpath(mod) if mod != "?" else mod
            if real not in tabs and os.path.exists(real):
                tabs[real] = symbols(real)
            b = base.get(mod, base.get(real, 0))
            # position-independent objects: nm's addresses are relative to the load base; the main program here is not PIE-relocated when built -no-pie
            off = addr - b if (addr - b) >= 0 and real in tabs and tabs[real] and tabs[real][-1][0] < 0x10000000 else addr
            print("   ", ln.split("[")[0].strip(), "=>", resolve(tabs.get(real, []), off), "(%s)" % os.path.basename(mod))
        else:
            print("   ", ln)


def step(cmd, cwd, limit, tabs, env_extra=None):
    dump_path = os.path.join(cwd, "stacks.txt")
    env = dict(os.environ, LD_PRELOAD=PROBE, RCGPU_HANG_DUMP=dump_path, **(env_extra or {}))
    t0 = time.time()
    p = subprocess.Popen(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, env=env, text=True)
    try:
        out, err = p.communicate(timeout=limit)
        return p.returncode, out, err, time.time() - t0
    except subprocess.TimeoutExpired:
        print("TIME-OUT after %.1f s: %s" % (limit, " ".join(cmd)))
        dump(p, cmd[0], dump_path, tabs)
        p.kill()
        out, err = p.communicate()
        print("  stdout tail:", out[-300:].replace("\n", " | "))
        print("  stderr tail:", err[-300:].replace("\n", " | "))
        sys.stdout.flush()
        return None, out, err, time.time() - t0


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    limit = float(sys.argv[2]) if len(sys.argv) > 2 else 20
    work = tempfile.mkdtemp(prefix="rcgpu_hang_")
    tabs = {}
    counts = {}
    try:
        os.makedirs(os.path.join(work, "pkg", "img"))
        w, h, pixfmt = 160, 90, synth.PIX_RGB16_BE
        for i in range(8):
            with open(os.path.join(work, "pkg", "img", "f_%06d.dpx" % i), "wb") as f:
                f.write(synth.dpx_file(synth.components(w, h, 3, 16, "film", seed=i), pixfmt, frame_index=i))
        with open(os.path.join(work, "pkg", "snd.wav"), "wb") as f:
            f.write(synth.wav_file(synth.pcm_samples(16000, 2, 16, 48000), 16, 48000))
        # RCGPU_HANG_BATCH=n: route C in batches of n frames (several batches of the 8 frames: one of them decoded ahead of its turn)
        benv = {"RCGPU_CHECK_BATCH": os.environ["RCGPU_HANG_BATCH"]} if os.environ.get("RCGPU_HANG_BATCH") else None
        steps = [
            ("ref -d (analysis, prints the ffmpeg command)", [REF, "--hash", "--check-padding", "-d", "-y", "pkg"], None),
            ("linked -d (routes D)", [LINKED, "--hash", "--no-check-padding", "-d", "-y", "pkg"], None),
            ("linked whole product (routes B, C, D)", [LINKED, "--no-check-padding", "--check", "--hash", "-y", "pkg"], benv),
            ("ref --check of the file", [REF, "--check", "pkg.mkv"], None),
            ("linked --check of the file (route C)", [LINKED, "--check", "pkg.mkv"], benv),
        ]
        t_all = time.time()
        for r in range(rounds):
            for name, cmd, env in steps:
                rc, out, err, dt = step(cmd, work, limit, tabs, env)
                c = counts.setdefault(name, {"runs": 0, "timeouts": 0, "failures": 0, "seconds": 0.0})
                c["runs"] += 1; c["seconds"] += dt
                if rc is None:
                    c["timeouts"] += 1
                elif rc != 0:
                    c["failures"] += 1
                    print("FAILED (%d): %s\n  %s" % (rc, " ".join(cmd), (out + err)[-400:].replace("\n", " | ")))
        print("==== %d rounds in %.0f s" % (rounds, time.time() - t_all))
        for name, c in counts.items():
            print("%-48s runs %4d  time-outs %3d  failures %3d  average %.2f s" % (name, c["runs"], c["timeouts"], c["failures"], c["seconds"] / max(1, c["runs"])))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
