#!/usr/bin/env python3
"""bench.py -- 4K-DCI 16-bit DPX -> FFV1 frames/s on MI355X (BASELINE.json metric, configs[1]).

One "step" = one pass of the whole device hot path (unpack+RCT, context model, state resolution, range coder,
footer/CRC, packet gather) over one batch of synthetic 4096x2160 RGB 16-bit big-endian DPX payloads that are
already resident in HBM, -slices 64 (8x8), -coder 1 -context 1 -slicecrc 1 -g 1 -level 3 -- the options RAWcooked
passes (Source/CLI/Global.cpp:938-989).  Frames shard across ranks with no collective: every rank
encodes its own batch; torch.distributed (RCCL) is used only for the barriers and the max-over-ranks timing.

The JSON line carries, beside the device-resident headline (`value`, the driver's contract):
  host_pipeline  the same workload with payloads starting and packets ending in HOST memory: reader threads fill pinned slots, every
                 batch is uploaded while the previous one is coded and downloaded while the next one is (rcgpu_ffv1_encode_sequence);
                 every rank runs it, so at N > 1 it shows where the GPUs contend (host memory, PCIe root)
  e2e            files on tmpfs -> rcgpu-ffmpeg (the argv the reference prints) -> MKV on tmpfs, process start to exit (N = 1)
  check          BASELINE config 5: device FFV1 decode + byte compare + MD5 of the same packets (N = 1)
  cpu_baseline   the scalar oracle on all host cores and on one (kind "port": the reference's encoder is FFmpeg, absent here)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch F] [--kind film|flat|noise] [--legs host,e2e,check,cpu]
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W4K, H4K = 4096, 2160
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


LINE_LIMIT = 4096             # the driver keeps the last 8 KB of stdout: the line it parses stays under half of that
DETAIL_FILE = "bench_detail.json"


def compact_line(result: dict, step_ms=None, detail_file: str = DETAIL_FILE) -> str:
    """The ONE line the driver parses: the contract's keys, `config` as a one-sentence workload plus flat numbers, `roofline` and
    `cpu_baseline` as numbers, the per-step times.  Everything else bench.py measures (per-leg records, traces, prose) stays in
    `result`, which main() writes to bench_detail.json (stderr gets one summary line: the driver's record joins the tails of both
    streams).  Never longer than LINE_LIMIT bytes: what does not fit is
    dropped from the END of `config` (the contract's keys come first)."""
    def scalar(v):
        return v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 48)
    src_cfg = result.get("config") or {}
    cfg = {"workload": str(src_cfg.get("workload", "")).split(";")[0][:200]}
    for k, v in src_cfg.items():
        if k != "workload" and scalar(v) and v is not None:
            cfg[k] = v
    r = result.get("roofline") or {}
    roof = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "request_frac", "floor_ms", "launch_ms",
                                  "launches_per_step", "algorithmic_bytes_per_launch")} if r else None
    if r and isinstance(r.get("kernel_ms_per_step"), dict):
        roof["kernel_ms_per_step"] = {k: round(v, 1) for k, v in r["kernel_ms_per_step"].items()}
    cb = result.get("cpu_baseline")
    cpu = None
    if isinstance(cb, dict) and "value" in cb:
        cpu = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}
        cpu["sample"] = str(cb.get("sample", "")).split(";")[0].split("(")[0].strip()[:120]
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                       "dtype", "data", "h2d_included")}
    km = result.get("kernel_metric")
    if isinstance(km, dict) and "value" in km:      # inputs in pinned host memory, H2D included
        line["kernel_metric"] = {"value": km["value"], "unit": km.get("unit", "frames/s"), "h2d_included": True, "frames": km.get("frames")}
    line["config"] = cfg
    line["roofline"] = roof
    line["cpu_baseline"] = cpu
    if step_ms:
        line["step_ms"] = [round(float(x), 1) for x in step_ms]
    line["detail"] = detail_file
    s = json.dumps(line, separators=(",", ":"))
    keys = [k for k in cfg if k != "workload"]
    while len(s) >= LINE_LIMIT and keys:
        cfg.pop(keys.pop())
        s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:      # (cannot happen with the keys above: a hundred step times would do it)
        line.pop("step_ms", None)
        s = json.dumps(line, separators=(",", ":"))
    return s


def make_frames(torch, n, width, height, kind, seed, device):
    """n synthetic RGB16-BE payloads on the device, uint8 [n, height*width*6]; generated 8 frames at a time."""
    g = torch.Generator(device=device)
    g.manual_seed(1000 + seed)
    maxv = 65535.0
    out = torch.empty((n, height * width * 6), dtype=torch.uint8, device=device)
    y = torch.linspace(0, 1, height, device=device).view(1, height, 1, 1)
    x = torch.linspace(0, 1, width, device=device).view(1, 1, width, 1)
    c = torch.arange(3, device=device).view(1, 1, 1, 3).float()
    for i0 in range(0, n, 8):
        m = min(8, n - i0)
        if kind == "flat":
            v = torch.randint(0, 65536, (m, 1, 1, 3), generator=g, device=device, dtype=torch.int32).expand(m, height, width, 3)
        elif kind == "noise":
            v = torch.randint(0, 65536, (m, height, width, 3), generator=g, device=device, dtype=torch.int32)
        else:
            # smooth picture + grain with sigma = 1/64 of full scale: the low ~10 bits are noise, like scanned film
            ph = torch.rand((m, 1, 1, 3), generator=g, device=device) * 6.28
            sig = 0.45 + 0.25 * torch.sin(3.1 * x + ph) * torch.cos(2.3 * y + ph * 0.7) + 0.15 * (x * (c + 1) / 3 + y * 0.5) \
                + 0.05 * torch.sin(40 * x + 31 * y + ph * 1.3)
            grain = torch.randn((m, height, width, 3), generator=g, device=device) / 64.0
            v = torch.clamp((sig + grain) * maxv * 0.8, 0, maxv).to(torch.int32)
        out[i0:i0 + m] = torch.stack(((v >> 8).to(torch.uint8), (v & 0xFF).to(torch.uint8)), dim=-1).reshape(m, height * width * 6)
        del v
    torch.cuda.empty_cache()
    return out


def usable_cores() -> int:
    """Hardware threads this process may really keep busy: the affinity mask, cut by the cgroup's CPU quota (the GPU boxes of this pool show
    256 threads and grant 16 CPUs' worth of time -- cpu.max "1600000 100000" --, which is why round 2's "256-thread" baselines scaled 13x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return max(1, n)


_CPU_WORKER = r"""
import hashlib, os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle_binding as ob
from rawcooked_amd import synth
width, height, line_bytes, nframes, go_at = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
payload = open(sys.argv[2], "rb").read()
p = ob.Params(width, height, synth.PIX_RGB16_BE, 8, 8, 1, 1)
ob.lib()
while time.time() < go_at:
    time.sleep(0.001)
t0 = time.time()
for _ in range(nframes):
    pk = ob.encode_payload(p, payload, line_bytes)
print(t0, time.time(), hashlib.md5(pk).hexdigest())
"""


def cpu_baseline(payloads: list, line_bytes: int, width: int, height: int, gpu_packet_md5: list):
    """The oracle (scalar C restatement, kind 'port') on the host.  One PROCESS per core -- round 2 ran it in threads of one process and
    got 13x on 256 threads: the oracle allocates ~0.7 GB per call, and one process's allocator and page-fault path serialised them.
    Worker k encodes distinct frame k % len(payloads), so the baseline doubles as a packet-by-packet check of the GPU's output
    (gpu_packet_md5[k]: MD5 of the device-resident run's packet of that frame)."""
    import hashlib
    import subprocess
    import tempfile
    cores = usable_cores()
    try:                                            # ~0.8 GB per worker at 4K: never let the baseline endanger the box
        import psutil
        cores = max(1, min(cores, int(psutil.virtual_memory().available / (1 << 30) / 2)))
    except Exception:
        cores = min(cores, 16)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    files = []
    for k, pl in enumerate(payloads):
        fn = os.path.join(base, "rcgpu_cpu_%d_%d.bin" % (os.getpid(), k))
        with open(fn, "wb") as f:
            f.write(pl)
        files.append(fn)

    def run(nproc, nframes):
        go_at = time.time() + 1.0 + nproc * 0.012            # every worker has loaded its input and the library before the clock starts
        procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, files[k % len(files)], str(width), str(height), str(line_bytes), str(nframes), repr(go_at)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(nproc)]
        rows = []
        for k, pr in enumerate(procs):
            out, err = pr.communicate(timeout=600)
            if pr.returncode != 0:
                raise RuntimeError("oracle worker failed: " + err[-300:])
            t0, t1, md5 = out.split()
            rows.append((float(t0), float(t1), md5, k % len(files)))
        span = max(r[1] for r in rows) - min(r[0] for r in rows)
        return nproc * nframes / span, rows

    try:
        one, rows1 = run(1, 1)
        allc, rows = run(cores, 2)
        checked = [(r[3], r[2]) for r in rows + rows1 if r[3] < len(gpu_packet_md5)]
        bad = sorted({k for k, md5 in checked if md5 != gpu_packet_md5[k]})
    finally:
        for fn in files:
            try:
                os.unlink(fn)
            except OSError:
                pass
    return {"value": round(allc, 4), "unit": "frames/s", "cores": cores, "cores_busy": cores, "hardware_threads_visible": os.cpu_count(), "kind": "port", "one_core": round(one, 4),
            "speedup_over_one_core": round(allc / one, 1),
            "gpu_packets_equal_to_oracle": {"frames_checked": len({k for k, _ in checked}), "differing": bad},
            "sample": f"{cores} processes x 2 frames of the same {width}x{height} RGB16 workload ({len(files)} distinct frames), started together, first start to last end "
                      f"(and 1 frame on 1 core: {one:.3f} frames/s); oracle/ffv1_oracle.c (scalar C, not FFmpeg; the reference's own encoder is the ffmpeg binary, "
                      f"absent here)"}, not bad


def linked_check_record(api, synth, record, frames, d_packets, stride, sizes, width, height, pixfmt, nframes=1000, cpu_frames=32, variants=None):
    """The product-level number of BASELINE config 5: `rawcooked_linked --check` -- the REAL reference with this library linked in
    (INTEGRATION.md route C: oracle/route_c_*.patch, built by oracle/Makefile.ref) -- on an MKV of this run's packets.  The demuxer
    announces the frames that follow, the device decoder takes them in batches (RCGPU_CHECK_BATCH) and keeps the payloads; frame_writer
    lets every rebuilt file wait for its batch, and the batch is hashed (MD5) and compared with the files on disk on the device
    (rcgpu_ffv1_decoder_verify_kept).  Beside it: the same binary with the payloads coming back to the reference's own per-file MD5 and
    memcmp (RCGPU_CHECK_DEFER=0), and with its own CPU slice pool (RCGPU_CHECK=0; a shorter MKV of the same packets, it is a steady
    per-frame rate).  Everything else is the reference's: demux, reversibility data, MergeIn, the verdict."""
    import shutil
    import subprocess
    import tempfile
    exe = os.environ.get("RCGPU_LINKED_EXE") or os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")       # e.g. a build with -pg
    if not os.path.exists(exe):
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    work = tempfile.mkdtemp(prefix="rcgpu_linked_", dir=base)
    try:
        F = len(sizes)
        n = max(1, nframes)
        payload_b = int(frames[0].numel())
        if base and shutil.disk_usage(base).free < 1.3 * (n * max(sizes) + min(n, F) * payload_b + (4 << 30)):
            n = min(n, 256)                       # BASELINE config 2 is 1000 frames; a box whose tmpfs cannot hold their MKV gets a shorter one, and says so
        m = min(n, max(1, cpu_frames))

        def run(cmd, cwd, env=None, timeout=90):
            # bounded, ONE retry: the lost wake-up of the reference's third-party thread pool (Lib/ThirdParty/thread-pool/include/ThreadPool.h:27-33
            # against :66-68) leaves matroska::Shutdown() in std::thread::join (Matroska.cpp:271-274); stacks in profiles/r04_hang_stacks.txt
            for attempt in range(2):
                try:
                    t0 = time.perf_counter()
                    r_ = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=timeout, env=env)
                    r_.seconds = time.perf_counter() - t0
                    return r_
                except subprocess.TimeoutExpired:
                    if attempt == 1:
                        raise

        def package(name, count):
            d = os.path.join(work, name)
            os.makedirs(os.path.join(d, "seq"))
            os.makedirs(os.path.join(d, "uniq"))
            host = {}
            for i in range(count):              # the sequence: F distinct pictures, the rest hard links to them (SURVEY.md 8d "ring reuse"; the e2e leg does the same)
                k = i % F
                if k not in host:
                    host[k] = bytes(d_packets[k * stride:k * stride + sizes[k]].cpu().numpy())
                    with open(os.path.join(d, "uniq", "u_%06d.dpx" % k), "wb") as f:
                        f.write(synth.dpx_file(None, pixfmt, frame_index=k, payload=bytes(frames[k].cpu().numpy()), size=(width, height)))
                os.link(os.path.join(d, "uniq", "u_%06d.dpx" % k), os.path.join(d, "seq", "f_%06d.dpx" % i))
            r = run([exe, "--hash", "--no-check-padding", "-d", "-y", "seq"], d, timeout=120)      # analysis only: the reversibility data with the MD5 of every file (route D)
            if r.returncode != 0:
                raise RuntimeError("analysis failed: " + (r.stderr or r.stdout)[-200:])
            mux = api.MkvMuxer(os.path.join(d, "seq.mkv"))
            t = mux.add_video(record, width, height, 24, 1)
            mux.add_attachment("RAWcooked reversibility data", open(os.path.join(d, "seq.rawcooked_reversibility_data"), "rb").read())
            mux.begin()
            for i in range(count):
                mux.write_block(t, i * 1000000000 // 24, host[i % F])
            mux.close()
            return d, r.seconds
        big, analysis_s = package("big", n)
        small = big if m == n else package("small", m)[0]
        out = {"frames": n, "unit": "frames/s", "analysis_seconds": round(analysis_s, 2)}
        OKL = "Reversibility was checked, no issue detected."
        # `--check x.mkv` judges by the MD5s in the reversibility data; with `-o .` the rebuilt files are also compared with the sources on disk
        # frames per device call: the patch's own choice (by slice chains: 768 frames for streams of up to 128 slices, else 256) unless RCGPU_LINKED_BATCH says otherwise
        benv = {"RCGPU_CHECK_BATCH": os.environ["RCGPU_LINKED_BATCH"]} if os.environ.get("RCGPU_LINKED_BATCH") else {}
        todo = (("device_decoder", dict({"RCGPU_CHECK": "1"}, **benv), big, n, []),
                ("device_decoder_and_sources", dict({"RCGPU_CHECK": "1"}, **benv), big, n, ["-o", "."]),
                ("device_decoder_payloads_to_host", {"RCGPU_CHECK": "1", "RCGPU_CHECK_DEFER": "0", "RCGPU_CHECK_BATCH": "128"}, small, m, []),
                ("reference_cpu_pool", {"RCGPU_CHECK": "0"}, small, m, []),
                ("reference_cpu_pool_and_sources", {"RCGPU_CHECK": "0"}, small, m, ["-o", "."]))
        for name, env, d, count, extra in todo:
            # by default: the device decoder and the reference's own pool; the other variants (sources compared too, payloads back to the host) on request
            # (RCGPU_LINKED_VARIANTS=device_decoder,device_decoder_and_sources,...): they were the subject of rounds 3-4 and cost the line 20 s
            if (variants and name not in variants) or (not variants and name not in ("device_decoder", "reference_cpu_pool")):
                continue
            # the process before this one has just given tens of GB of device memory back, which the driver wipes in the background: an allocation
            # that follows within ~3 s waits for the wipe (measured: 3 s for 39 GB that take 0.1 s on an idle device).  A job does not follow another
            # one's exit by milliseconds, so the wipe is allowed to finish before the clock starts (the e2e leg does the same).
            time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
            r = run([exe, "--check", "seq.mkv"] + extra, d, env=dict(os.environ, **env), timeout=180)
            ok = r.returncode == 0 and OKL in r.stdout
            out[name] = {"value": round(count / r.seconds, 2), "frames": count, "seconds": round(r.seconds, 2), "verdict": OKL if ok else (r.stdout + r.stderr)[-200:]}
            if not ok:
                out[name]["returncode"] = r.returncode
                print("bench: %s: exit status %s\n%s" % (name, r.returncode, "\n".join(ln for ln in (r.stdout + r.stderr).replace("\r", "\n").split("\n") if ln.strip() and "Time=" not in ln and "rcgpu kept" not in ln)[-1500:]), file=sys.stderr)
            if os.environ.get("RCGPU_TRACE_KEPT"):
                import resource
                ru = resource.getrusage(resource.RUSAGE_CHILDREN)
                out[name]["children_cpu_seconds_so_far"] = {"user": round(ru.ru_utime, 2), "system": round(ru.ru_stime, 2), "minor_faults": ru.ru_minflt}
                out[name]["trace"] = [ln[ln.index("rcgpu "):] for ln in r.stderr.replace("\r", "\n").split("\n") if "rcgpu " in ln]
        if not variants or "whole_product" in variants:
            # and the product as a user runs it: one process analyses the sources (route D), encodes them (route B), and checks the MKV it wrote (route C)
            os.rename(os.path.join(big, "seq.mkv"), os.path.join(big, "muxed_by_the_bench.mkv"))
            time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
            r = run([exe, "--no-check-padding", "--check", "--hash", "-y", "seq"], big, env=dict(os.environ, RCGPU_CHECK="1", **benv), timeout=300)
            ok = r.returncode == 0 and OKL in r.stdout
            same = ok and os.path.getsize(os.path.join(big, "seq.mkv")) > 0
            if os.environ.get("RCGPU_TRACE_KEPT"):
                out["whole_product_trace"] = [ln[ln.index("rcgpu"):] for ln in r.stderr.replace("\r", "\n").split("\n") if "rcgpu" in ln][:60]
            out["whole_product"] = {"value": round(n / r.seconds, 2), "frames": n, "seconds": round(r.seconds, 2), "verdict": OKL if ok else (r.stdout + r.stderr)[-200:],
                                    "what": "rawcooked_linked --check --hash -y <folder>: analysis with the MD5 of every source, FFV1 encoding into a Matroska file, and the check of that file, "
                                            "process start to exit", "mkv_bytes": os.path.getsize(os.path.join(big, "seq.mkv")) if same else 0}
        if out.get("reference_cpu_pool", {}).get("value") and "device_decoder" in out:
            out["speedup"] = round(out["device_decoder"]["value"] / out["reference_cpu_pool"]["value"], 2)
        out["what"] = (f"oracle/_ref/rawcooked_linked --check on an MKV of {n} {width}x{height} frames (this run's packets, tmpfs), process start to exit: the reference's own demuxer, "
                       f"reversibility data and verdict around the device decoder, the files hashed and compared with the sources on the device a batch at a time; "
                       f"reference_cpu_pool = the same binary with RCGPU_CHECK=0 on {m} of the frames; {usable_cores()} usable cores")
        return out
    except Exception as e:
        return {"error": str(e)[-300:]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def long_sequence_leg(synth, host_ring, width, height, n_frames, pixfmt):
    """The product on a sequence long enough to show its steady state (real sequences are 7 000 frames and more, Doc/Case_study.md:234-238;
    the 1000-frame legs are three or four batches, mostly fill and drain): n_frames DPX files on tmpfs (the ring's distinct pictures,
    hard-linked), then two processes of the REAL reference with this library linked in (oracle/Makefile.ref `linked`), each timed from
    start to exit:  `rawcooked_linked --hash --no-check -y seq` -- analysis with the MD5 of every source (route D), FFV1 encoding at the reference's own
    slice count into one Matroska file (route B) -- and `rawcooked_linked --check seq.mkv` (route C: the device decoder in batches, the
    rebuilt files hashed on the device).  Steady rates from the processes' own traces: batches after the first / the time between the
    first and the last batch's completion."""
    import re
    import shutil
    import struct
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")
    if not os.path.exists(exe):
        return {"skipped": "oracle/_ref/rawcooked_linked is not here"}, True
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    R = len(host_ring)
    payload = host_ring[0].nbytes
    if not base or shutil.disk_usage(base).free < 1.2 * (n_frames * payload + R * payload) + (8 << 30):
        return {"skipped": "no tmpfs with room for a %d-frame Matroska file" % n_frames}, True
    work = tempfile.mkdtemp(prefix="rcgpu_long_", dir=base)
    OKL = "Reversibility was checked, no issue detected."
    try:
        os.makedirs(os.path.join(work, "uniq")); os.makedirs(os.path.join(work, "seq"))
        hdr = synth.dpx_file(np.zeros((1, 1, 3), dtype=np.uint16), pixfmt)[:2048]
        for i in range(R):
            h = bytearray(hdr)
            struct.pack_into(">I", h, 772, width); struct.pack_into(">I", h, 776, height); struct.pack_into(">I", h, 16, 2048 + payload)
            with open(os.path.join(work, "uniq", "u_%03d.dpx" % i), "wb") as f:
                f.write(h); f.write(memoryview(host_ring[i]))
        for i in range(n_frames):
            os.link(os.path.join(work, "uniq", "u_%03d.dpx" % (i % R)), os.path.join(work, "seq", "f_%06d.dpx" % i))
        env = dict(os.environ, RCGPU_TRACE="1", RCGPU_TRACE_KEPT="1", RCGPU_CHECK="1")
        out = {"frames": n_frames, "unit": "frames/s", "distinct_pictures": R}

        def run(cmd, timeout):
            # stderr is read as it comes, every line with its arrival time: the processes' own traces say WHERE a long run spends its time
            import threading
            for attempt in range(2):      # one retry: the lost shutdown wake-up of the reference's thread pool (profiles/r04_hang_stacks.txt)
                t0 = time.perf_counter()
                pr = subprocess.Popen(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, text=True, env=env)
                lines, outbuf = [], []

                def pump_err():
                    for ln in pr.stderr:
                        for part in ln.replace("\r", "\n").split("\n"):
                            if part.strip():
                                lines.append((round(time.perf_counter() - t0, 2), part.rstrip()))
                th = threading.Thread(target=pump_err); th.start()
                th2 = threading.Thread(target=lambda: outbuf.append(pr.stdout.read())); th2.start()
                try:
                    pr.wait(timeout=timeout)
                except subprocess.TimeoutExpired:
                    pr.kill(); pr.wait()
                    if attempt == 1:
                        raise
                    continue
                th.join(); th2.join()
                class R_: pass
                r_ = R_()
                r_.returncode = pr.returncode; r_.stdout = outbuf[0] if outbuf else ""; r_.stderr = "\n".join(l for _, l in lines); r_.seconds = time.perf_counter() - t0
                r_.timeline = ["%7.2f s  %s" % (t, l[:160]) for t, l in lines if "rcgpu" in l and "rcgpu kept" not in l]
                return r_
        time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
        # --no-check: without it the reference parses the file it has just written once more and, because of --hash, MD5s the whole Matroska file on
        # one core (input_base::Parse -> Hash, Lib/Utils/FileIO/Input_Base.cpp:38-81: 129 s for the 119 GB of 2000 frames); the check is the second process
        r = run([exe, "--hash", "--no-check-padding", "--no-check", "-y", "seq"], 600)
        ok_e = r.returncode == 0 and os.path.exists(os.path.join(work, "seq.mkv"))
        slices = re.search(r"-slices (\d+)", r.stdout + r.stderr)
        pl = [ln for ln in r.stderr.splitlines() if "pipeline:" in ln and "frames in" in ln]
        steady = re.search(r"frames/s; ([0-9.]+) between the first and the last batch", pl[-1]) if pl else None
        out["analysis_and_encode"] = {"value": round(n_frames / r.seconds, 2), "seconds": round(r.seconds, 2), "slices": int(slices.group(1)) if slices else None,
                                      "mkv_bytes": os.path.getsize(os.path.join(work, "seq.mkv")) if ok_e else 0,
                                      "encode_steady_frames_per_second": float(steady.group(1)) if steady else None,
                                      "trace": pl[-1].split("pipeline: ", 1)[1][:400] if pl else None, "timeline": r.timeline[:12] + (["..."] if len(r.timeline) > 40 else []) + r.timeline[12:][-28:],
                                      **({} if ok_e else {"error": (r.stdout + r.stderr)[-300:]})}
        ok_c = False
        if ok_e:
            time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
            r = run([exe, "--check", "seq.mkv"], 600)
            ok_c = r.returncode == 0 and OKL in r.stdout
            done = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"verify_kept: md5 done\s+(\d+) files\s+[0-9.]+ ms\s+\(at ([0-9.]+) s\)", r.stderr)]
            steady_c = None
            if len(done) >= 3 and done[-1][1] > done[0][1]:
                steady_c = round(sum(n for n, _ in done[1:]) / (done[-1][1] - done[0][1]), 2)
            out["linked_check"] = {"value": round(n_frames / r.seconds, 2), "seconds": round(r.seconds, 2), "verdict": OKL if ok_c else (r.stdout + r.stderr)[-200:],
                                   "batches": len(done), "steady_frames_per_second": steady_c,
                                   "first_batch_verified_at_s": done[0][1] if done else None, "last_batch_verified_at_s": done[-1][1] if done else None}
        out["what"] = (f"{n_frames} x {width}x{height} RGB16 DPX on tmpfs ({R} distinct, hard-linked); rawcooked_linked --hash --no-check -y (analysis + encoding at the reference's own slice count, one MKV) "
                       f"and rawcooked_linked --check of that file, each process start to exit; steady = batches after the first over the time between the first and the last batch's completion")
        return out, ok_e and ok_c
    except Exception as e:
        return {"error": str(e)[-300:]}, False
    finally:
        shutil.rmtree(work, ignore_errors=True)


def reference_check_baseline(api, synth, record, frames, d_packets, stride, sizes, width, height, pixfmt, nframes=16, parallel=1):
    """cpu_baseline of the check path, kind "reference": the REAL reference binary (oracle/_ref/rawcooked, built from the reference's
    own sources by oracle/Makefile.ref) decodes and verifies an MKV holding `nframes` of this run's packets, on the host's cores."""
    import shutil
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
    if not os.path.exists(ref):
        return None
    work = tempfile.mkdtemp(prefix="rcgpu_bench_")
    try:
        os.makedirs(os.path.join(work, "seq"))
        n = min(nframes, len(sizes))
        for i in range(n):
            with open(os.path.join(work, "seq", "f_%06d.dpx" % i), "wb") as f:
                f.write(synth.dpx_file(None, pixfmt, frame_index=i, payload=bytes(frames[i].cpu().numpy()), size=(width, height)))
        def run(cmd):       # bounded, one retry: the reference's thread pool can lose its shutdown wake-up (ThreadPool.h:27-33,66-68; profiles/r04_hang_stacks.txt)
            for attempt in (0, 1):
                try:
                    return subprocess.run(cmd, cwd=work, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60)
                except subprocess.TimeoutExpired:
                    if attempt:
                        raise
        r = run([ref, "--hash", "--no-check-padding", "-d", "-y", "seq"])     # analysis only: writes the reversibility data with the MD5 of every file
        if r.returncode != 0:
            return None
        mux = api.MkvMuxer(os.path.join(work, "seq.mkv"))
        t = mux.add_video(record, width, height, 24, 1)
        mux.add_attachment("RAWcooked reversibility data", open(os.path.join(work, "seq.rawcooked_reversibility_data"), "rb").read())
        mux.begin()
        for i in range(n):
            pkt = bytes(d_packets[i * stride:i * stride + sizes[i]].cpu().numpy())
            mux.write_block(t, i * 1000000000 // 24, pkt)
        mux.close()
        t0 = time.perf_counter()
        r = run([ref, "--check", "seq.mkv"])
        dt = time.perf_counter() - t0
        ok = r.returncode == 0 and "Reversibility was checked, no issue detected." in r.stdout
        rec = {"value": round(n / dt, 3), "unit": "frames/s", "cores": usable_cores(), "hardware_threads_visible": os.cpu_count(), "kind": "reference",
               "sample": f"rawcooked --check on an MKV of {n} of this run's {width}x{height} packets (files on local disk, page cache warm); "
                         f"verdict: {'no issue detected' if ok else 'FAILED'}"}
        if parallel > 1 and ok:
            # the reference decodes one frame at a time on a pool of 64 slice tasks: one process cannot fill a 256-thread host.  Its own
            # documentation runs packages side by side with GNU parallel (Doc/Case_study.md:81); so does this.
            t0 = time.perf_counter()
            procs = [subprocess.Popen([ref, "--check", "seq.mkv"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, text=True)
                     for _ in range(parallel)]
            good = 0
            for pr in procs:
                try:
                    out, _ = pr.communicate(timeout=120)
                    good += pr.returncode == 0 and "Reversibility was checked, no issue detected." in out
                except subprocess.TimeoutExpired:
                    pr.kill()
            dtp = time.perf_counter() - t0
            if good == parallel:
                rec.update({"one_process": rec["value"], "value": round(parallel * n / dtp, 3), "processes": parallel,
                            "cores_busy": min(usable_cores(), parallel * 64)})
                rec["sample"] += f"; value = {parallel} such processes side by side ({n} frames each, {dtp:.1f} s), one process alone {rec['one_process']} frames/s"
        return rec
    except Exception as e:      # the baseline is a report, never a reason to fail the measurement
        print("bench: reference check baseline skipped:", e, file=sys.stderr)
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


_TRAFFIC = None


def traffic_json():
    """profiles/traffic.json, and whether it still describes the code: it records the sha256 of the kernel sources the PMC passes ran on
    (tools/update_traffic.py); when ffv1_gpu.hip / ffv1_check.hip have changed since, the byte counts are withheld (`traffic: null,
    traffic_stale: true`) instead of being printed beside kernels they were not measured on."""
    global _TRAFFIC
    if _TRAFFIC is None:
        import hashlib
        d, stale = {}, {}
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            # ... or, the sources having changed, the sha256 of the library's DEVICE code (rawcooked_amd/devcode.py): host code edited inside a
            # .hip file leaves the kernels what they were
            from rawcooked_amd import devcode
            same_kernels = bool(d.get("device_code_sha256")) and devcode.fatbin_sha256(os.path.join(ROOT, "rawcooked_amd", "librcgpu.so")) == d["device_code_sha256"]
            for name, want in d.get("sources", {}).items():
                got = hashlib.sha256(open(os.path.join(ROOT, name), "rb").read()).hexdigest()
                stale[os.path.basename(name)] = got != want and not same_kernels
        except Exception:
            d = {}
        _TRAFFIC = (d, stale)
    return _TRAFFIC


def traffic_of(kernel, source):
    """(HBM bytes per 4K frame of `kernel` from the PMC passes or None, stale?) -- source = the file the kernel lives in"""
    d, stale = traffic_json()
    if not d.get("sources") or stale.get(source, True):
        return None, True
    return d.get(kernel, {}).get("per_frame_bytes"), False


def request_roofline(records_per_second):
    """The roofline that binds the check half: one random 32-byte state record gathered and written back per sample, against what
    tools/gather_region measured the chip to sustain in the decoder's pattern -- every lane in a state array of its own
    (profiles/traffic.json: request_ceiling)."""
    try:
        rc_ = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["request_ceiling"]
        g = records_per_second / 1e9
        return {"request_frac": round(g / rc_["decoder_G_records_per_s"], 4),
                "requests": {"records_G_per_s": round(g, 2), "ceiling_G_per_s": rc_["decoder_G_records_per_s"], "uniform_address_ceiling_G_per_s": rc_["uniform_G_records_per_s"],
                             "ceiling_what": rc_["what"]}}
    except Exception:
        return {}


def check_leg(args, torch, api, record, frames, d_packets, sizes, ptrs, stride, stream, F, width, height, line_bytes, nh, nv, pixfmt, device, steps, warmup, cpu=True,
              check_batch=None):
    """Config 5: decode the packets back and verify them -- everything resident in HBM (single GPU).  The caller has released the
    encoder: the decoder is latency-bound per slice chain, its rate is chains in flight / chain latency, so D >= F frames are decoded per
    step (the F encoded packets, reused round-robin -- SURVEY.md 8d "ring reuse")."""
    cpu_rec = linked_rec = None
    if cpu:
        from rawcooked_amd import synth as _synth
        only = set(filter(None, os.environ.get("RCGPU_LINKED_VARIANTS", "").split(","))) or None      # (measuring: some of the variants only)
        linked_rec = linked_check_record(api, _synth, record, frames, d_packets, stride, sizes, width, height, pixfmt, variants=only)
        cpu_rec = reference_check_baseline(api, _synth, record, frames, d_packets, stride, sizes, width, height, pixfmt, parallel=max(1, min(8, usable_cores() // 32)))
    torch.cuda.empty_cache()
    D = max(F, check_batch or args.check_batch)
    sizes = [sizes[i % F] for i in range(D)]
    dec = api.Ffv1Decoder(width, height, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=D, device=device)
    payload_bytes = line_bytes * height
    # Two sets of output buffers when they fit: while batch k is decoded, the MD5 of every frame of batch k-1 is computed on a second
    # stream (one lane per frame: ~1.1 s for a 53 MB frame whatever the count, hidden behind the 3.7 s of decoding) and compared
    # with hashlib's digest of the source -- FileWriter.cpp:596-727's verification, inside the timed region.
    outs = [torch.empty((D, payload_bytes), dtype=torch.uint8, device=frames.device)]
    try:
        outs.append(torch.empty((D, payload_bytes), dtype=torch.uint8, device=frames.device))
    except RuntimeError:
        torch.cuda.empty_cache()
    pipelined = len(outs) == 2
    pk = [d_packets.data_ptr() + (i % F) * stride for i in range(D)]
    ops = [[o[i].data_ptr() for i in range(D)] for o in outs]
    ptrs = [ptrs[i % F] for i in range(D)]
    import hashlib
    F0 = min(F, 32)                                                                        # distinct sources hashed on the host
    want = {i: hashlib.md5(bytes(frames[i].cpu().numpy())).digest() for i in range(F0)}
    side = torch.cuda.Stream()
    # The decoding gets a stream of its own too, not torch's current one: that is the legacy default stream, with which every blocking
    # stream synchronises -- and the library's CU-masked streams (the hash on CUs of its own, ffv1_check.hip partition_streams) can only be
    # made blocking: a hash launched while the default stream waits for the decoder would itself wait for the decoder.
    main = torch.cuda.Stream()
    stream = main.cuda_stream
    state = {"k": 0, "bad": 0, "hashed": 0, "compared": 0, "differing": 0, "ev": None}
    spans = {"decode": [], "compare": [], "md5": []}          # HIP events around every launch: (begin, end) on the stream it was issued to

    def verify(prev, st, ts):
        # frame_writer's two checks (FileWriter.cpp:448-463 byte compare with the source, :596-727 MD5), on the device
        a0, a1, a2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a0.record(ts)
        diffs = api.compare_device_batch(prev, ptrs, [payload_bytes] * D, st)
        a1.record(ts)
        state["differing"] += sum(d != -1 for d in diffs); state["compared"] += D
        got = api.md5_device(prev, [payload_bytes] * D, st)
        a2.record(ts)
        spans["compare"].append((a0, a1)); spans["md5"].append((a1, a2))
        state["bad"] += sum(got[i] != want[i % F] for i in range(D) if (i % F) in want); state["hashed"] += D

    def step():
        k = state["k"]; cur = k & 1 if pipelined else 0
        e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
        dec.decode_device(pk, sizes, ops[cur], stream, check=False)
        ev = torch.cuda.Event(enable_timing=True); ev.record(main)
        spans["decode"].append((e0, ev))
        if pipelined and state["ev"] is not None:         # batch k-1 is verified on the side stream while batch k is decoded
            side.wait_event(state["ev"])
            verify(ops[cur ^ 1], side.cuda_stream, side)
        elif not pipelined:
            verify(ops[0], stream, main)
        state["ev"] = ev; state["k"] = k + 1

    # untimed warm-up steps until two in a row take the same time (3 %; four more at most): the legs before this one end by giving back
    # hundreds of GB, which the driver wipes in the background, and a decoder that runs beside that is not what is measured here
    last = None
    for w_ in range(max(0, warmup) + 4):
        torch.cuda.synchronize(); tw = time.perf_counter()
        step()
        torch.cuda.synchronize(); tw = time.perf_counter() - tw
        if w_ + 1 >= max(0, warmup) and last is not None and abs(tw - last) <= 0.03 * last:
            break
        last = tw
    torch.cuda.synchronize()
    for v_ in spans.values():
        v_.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # every launch of the timed steps by its own HIP events, in the configuration that was timed (the hash on CUs of its own: rocprofv3's
    # kernel trace cannot look at that, tools/rocprof_cumask_repro.hip) -- decode = k_dec_split + k_dec_crc + k_dec_slices of one batch
    per_launch = {k_: [round(a.elapsed_time(b), 2) for a, b in v_] for k_, v_ in spans.items()}
    op = ops[(state["k"] - 1) & 1 if pipelined else 0]                                     # the batch decoded last: verified below
    kt = dec.kernel_times()
    t1 = time.perf_counter()
    same = all(d == -1 for d in api.compare_device_batch(op, ptrs, [payload_bytes] * D, stream)) and state["differing"] == 0     # the last batch, outside the clock
    md5 = api.md5_device(op[:min(D, 64)], [line_bytes * height] * min(D, 64), stream)
    t_verify = time.perf_counter() - t1
    ok_md5 = md5[0] == want[0] and state["bad"] == 0
    payload = line_bytes * height
    packet_avg = sum(sizes) / len(sizes)
    dom = "k_dec_slices"
    achieved = D * (packet_avg + 2 * payload) / (kt[dom] * 1e-3) / 1e9          # packet in + payload out + the source again for the compare
    per_frame, traffic_stale = traffic_of(dom, "ffv1_check.hip")
    traffic = int(per_frame * D * (width * height) / (W4K * H4K)) if per_frame else None
    rec = {
        "metric": "4K-DCI 16-bit FFV1->DPX check frames/sec", "value": round(D * steps / dt, 3), "unit": "frames/s", "n_gpus": 1,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": f"{width}x{height} RGB16 FFV1 packets (slices={nh * nv}) -> payload, byte compare with the source + MD5 of every frame on device", "frames_per_step_per_gpu": D,
                   "all_frames_identical_to_source": bool(same), "md5_matches_hashlib": bool(ok_md5), "md5_inside_timed_region": state["hashed"],
                   "compared_inside_timed_region": state["compared"], "verify_seconds_last_batch": round(t_verify, 3)},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                     "traffic": traffic, "traffic_stale": traffic_stale, "kernel_ms": {k: round(v, 3) for k, v in kt.items()},
                     "per_launch_ms": per_launch, "hash_on_cus_of_its_own": not os.environ.get("RCGPU_NO_CU_PARTITION"), **request_roofline(width * height * 3 * D / (kt[dom] * 1e-3) if dom in kt and kt[dom] else 0)},
        **({"cpu_baseline": cpu_rec} if cpu_rec else {}), **({"linked_check": linked_rec} if linked_rec else {})}
    if getattr(args, "check_offsets", ""):
        # One allocation, several base addresses (rcgpu_ffv1_decoder_debug_states_offset; the decoder leaves room for them when RCGPU_DEC_STATES_SLACK=1
        # is set before it is made -- main() does for --check-offsets): does k_dec_slices' time depend on where its states lie?
        # Every offset in turn, three rounds, the kernel alone on the device (no hash beside it), its own HIP events.
        offs = [int(x) for x in args.check_offsets.split(",") if x.strip()]
        table = {str(o): [] for o in offs}
        for _ in range(3):
            for o in offs:
                dec.debug_states_offset(o)
                torch.cuda.synchronize()
                dec.decode_device(pk, sizes, ops[0], stream, check=False)
                torch.cuda.synchronize()
                table[str(o)].append(round(dec.kernel_times()["k_dec_slices"], 1))
        dec.debug_states_offset(0)
        rec["roofline"]["k_dec_slices_ms_by_states_offset"] = table
        # ... and on where the ALLOCATION lies: the decoder made anew three times in this process, another buffer allocated in between each time
        # (an offset moves virtual addresses inside one mapping; a new allocation gets other physical pages)
        alloc = []
        for gb in (0, 3, 11):
            dec.close()
            pad = torch.empty(gb << 30, dtype=torch.uint8, device=frames.device) if gb else None
            dec = api.Ffv1Decoder(width, height, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=D, device=device)
            ms = []
            for _ in range(2):
                torch.cuda.synchronize()
                dec.decode_device(pk, sizes, ops[0], stream, check=False)
                torch.cuda.synchronize()
                ms.append(round(dec.kernel_times()["k_dec_slices"], 1))
            alloc.append({"buffer_allocated_before_the_decoder_gb": gb, "k_dec_slices_ms": ms})
            del pad
        rec["roofline"]["k_dec_slices_ms_by_allocation"] = alloc
    dec.close()
    del outs
    torch.cuda.empty_cache()
    return rec, bool(same and ok_md5)


def host_pipeline_leg(api, cfg, host_ring, n_frames, batch, expect_packets, barrier, reduce_max, lanes=1, readers=0, writers=0, slots=0,
                      device_first=None, device_count=1, aliases=0, pinned=False):
    """The workload again with payloads starting and packets ending in host memory (pageable, like the page cache the reference's
    mmaps read from): rcgpu_ffv1_encode_sequence -- reader threads copy into pinned slots, upload / code / download overlap, writer
    threads copy every packet out of the pinned ring into host buffers.  expect_packets: device-resident packets of the ring's
    frames, byte-compared with what arrives here."""
    import ctypes as C
    import numpy as np
    R = len(host_ring)
    payload = host_ring[0].nbytes
    if pinned:
        # SURVEY.md 8d to the letter: "inputs resident in pinned host memory, H2D included" -- the ring is page-locked and the pipeline uploads
        # from it (rcgpu_sequence_options::frames_pinned): no reader threads, no copy into upload slots
        import torch
        pins = [torch.from_numpy(a).pin_memory() for a in host_ring]
        addrs = [t.data_ptr() for t in pins]
    else:
        addrs = [a.ctypes.data for a in host_ring]
    nout = 64
    out_cap = int(payload * 1.3) + (1 << 20)
    outs = [np.empty(out_cap, dtype=np.uint8) for _ in range(nout)]       # where packets end: pageable host memory, reused
    # batches of equal size (what the job level does by itself, pipeline.hip: 3840 frames are 12 x 320, not 11 x 336 + 144 -- a short last
    # batch costs a whole range-coder chain for a third of the frames)
    per_lane = -(-n_frames // max(1, device_count))
    nb = -(-per_lane // max(1, batch))
    batch = -(-per_lane // nb)
    barrier()
    t0 = time.perf_counter()
    st, sizes = api.encode_sequence_memory(cfg, addrs, n_frames, [a.ctypes.data for a in outs], out_cap, batch=batch // lanes,
                                           device_first=cfg.device if device_first is None else device_first, device_count=device_count, lanes_per_device=lanes,
                                           readers=readers, writers=writers, in_ring_frames=slots, device_aliases=aliases, frames_pinned=1 if pinned else 0)
    wall = time.perf_counter() - t0
    dt = reduce_max(st.seconds)          # the pipeline's own clock: first read to last packet, encoder creation (prepare_seconds) beside it
    # the last `nout` packets are still in their buffers: byte-compare them with the device-resident run's packets of the same frames
    bad = []
    checked = 0
    for f in range(max(0, n_frames - nout), n_frames):
        if f % R < len(expect_packets):
            checked += 1
            if bytes(outs[f % nout][:sizes[f]]) != expect_packets[f % R]:
                bad.append(f)
    check_until = checked
    return {"frames_per_gpu": n_frames, "seconds": round(dt, 3), "prepare_seconds": round(st.prepare_seconds, 3),
            "call_seconds": round(wall, 3), "first_packet_seconds": round(st.first_packet_seconds, 3),
            "h2d_GBps": round(st.payload_bytes / st.seconds / 1e9, 2), "d2h_GBps": round(st.packet_bytes / st.seconds / 1e9, 2),
            "batch_frames": st.batch_frames, "batches": st.batches, "readers": st.readers, "writers": st.writers,
            "steady_frames_per_second": round(st.steady_frames_per_second, 2),
            "reads_done_seconds": round(st.reads_done_seconds, 3), "last_batch_seconds": round(st.last_batch_seconds, 3),
            "upload_wait_seconds": round(st.upload_wait_seconds, 3), "h2d_span_seconds": round(st.h2d_span_seconds, 3),
            "read_call_ms": round(st.read_call_seconds * 1e3, 2), "write_call_ms": round(st.write_call_seconds * 1e3, 2),
            "packets_identical_to_device_resident_run": (not bad) if check_until else None,
            "lanes": [{"device": st.lane_device[i], "numa_node": st.lane_numa_node[i], "pinned_ring_on_node": st.lane_pinned_node[i]} for i in range(min(16, st.devices))],
            "host_groups": st.host_groups,
            "source": f"ring of {R} distinct frames in {'pinned' if pinned else 'pageable'} host memory, reused", "_local": (n_frames, st.seconds, dt)}, (not bad)


def e2e_leg(synth, host_ring, width, height, n_frames, slices, expect_packets, with_audio=True, device=None, tag="", batch=0):
    """Files on tmpfs -> `rcgpu-ffmpeg` with the argv grammar the reference assembles (Source/CLI/Output.cpp:81-310) -> MKV on tmpfs:
    the disk is out of the number, everything else (process start, device init, buffers, readers, muxer) is in it."""
    import shutil
    import struct
    import subprocess
    import numpy as np
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    work = os.path.join(base, "rcgpu_e2e_%d%s" % (os.getpid(), tag))
    shutil.rmtree(work, ignore_errors=True)
    try:
        os.makedirs(os.path.join(work, "uniq"))
        os.makedirs(os.path.join(work, "img"))
        R = len(host_ring)
        payload = host_ring[0].nbytes
        need = R * payload + int(n_frames * payload * 1.05)
        if shutil.disk_usage(base).free < need * 1.2:
            return {"skipped": f"{base} has no room for {need >> 30} GiB"}, True
        hdr = synth.dpx_file(np.zeros((1, 1, 3), dtype=np.uint16), synth.PIX_RGB16_BE)[:2048]
        for i in range(R):
            h = bytearray(hdr)
            struct.pack_into(">I", h, 772, width); struct.pack_into(">I", h, 776, height); struct.pack_into(">I", h, 16, 2048 + payload)
            with open(os.path.join(work, "uniq", "u_%03d.dpx" % i), "wb") as f:
                f.write(h); f.write(memoryview(host_ring[i]))
        for i in range(n_frames):                           # the sequence: hard links to the ring's files ("ring reuse")
            os.link(os.path.join(work, "uniq", "u_%03d.dpx" % (i % R)), os.path.join(work, "img", "f_%06d.dpx" % i))
        shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
        argv = [shim, "-xerror", "-framerate", "24.000000", "-r", "24.000000", "-f", "image2", "-c:v", "dpx", "-start_number", "000000",
                "-i", "img/f_%06d.dpx"]
        audio_note = ""
        if with_audio:      # BASELINE config 3: 6 ch / 24 bit / 48 kHz for the length of the sequence, the argv the reference prints for such a package
            nsamp = n_frames * 48000 // 24
            with open(os.path.join(work, "snd.wav"), "wb") as f:
                f.write(synth.wav_file(synth.pcm_samples(nsamp, 6, 24, 48000), 24, 48000))
            argv += ["-i", "snd.wav", "-map", "0", "-map", "1", "-c:a", "flac"]
            audio_note = f" + snd.wav (6 ch / 24 bit / 48 kHz, {nsamp} samples per channel) -> FLAC"
        argv += ["-c:v", "ffv1", "-coder", "1", "-context", "1", "-f", "matroska", "-g", "1", "-level", "3",
                 "-slicecrc", "1", "-slices", str(slices), "-y"] + (["-rcgpu_batch", str(batch)] if batch else []) + ["-f", "matroska", "out.mkv"]
        env = dict(os.environ, RCGPU_TRACE="1")
        if device is not None:
            env["RCGPU_DEVICES"] = "%d,1" % device          # this job's GPU (the shim's device selection; default: all visible)
        # The legs before this one have just given ~140 GB of device memory back, which the driver wipes in the background: a hipMalloc that
        # comes within ~3 s waits for the wipe (measured: encoder creation 3.2 s instead of 0.1 s).  A job does not follow another one's exit
        # by milliseconds, so the wipe is allowed to finish before the clock starts.
        time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
        t0 = time.perf_counter()
        r = subprocess.run(argv, cwd=work, capture_output=True, text=True, env=env, timeout=600)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}, False
        size = os.path.getsize(os.path.join(work, "out.mkv"))
        ok = None
        blocks = {"video": 0, "audio": 0, "differing": []}
        if expect_packets:
            # every SimpleBlock of the video track against the device-resident run's packet of that frame (the file is walked by the
            # structural validator of the tests, the payloads are compared through a mapping: no copy)
            import mmap
            import mkv_validator
            with open(os.path.join(work, "out.mkv"), "rb") as f:
                mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                view = memoryview(mm)
                expv = [memoryview(e) for e in expect_packets]

                def on_block(trk, t_abs, a, b):
                    if trk != 1:
                        blocks["audio"] += 1
                        return
                    i = blocks["video"]; blocks["video"] += 1
                    e = expv[i % R]
                    if b - a != len(e) or view[a:b] != e:
                        blocks["differing"].append(i)
                rep_ = mkv_validator.validate(os.path.join(work, "out.mkv"), on_block=on_block)
                del view, expv
                mm.close()
            ok = blocks["video"] == n_frames and not blocks["differing"] and rep_["tracks"][1]["codec"] == "V_FFV1"
        pl = [ln for ln in r.stderr.splitlines() if "pipeline:" in ln and "frames in" in ln]
        return {"frames": n_frames, "seconds": round(dt, 3), "value": round(n_frames / dt, 2), "unit": "frames/s", "mkv_bytes": size,
                "read_GBps": round(n_frames * (payload + 2048) / dt / 1e9, 2), "write_GBps": round(size / dt / 1e9, 2),
                "all_blocks_identical_to_device_resident_run": ok, "video_blocks": blocks["video"], "audio_blocks": blocks["audio"], "differing_blocks": blocks["differing"][:8],
                "single_output_file_ceiling": "one job writes one Matroska file: its page allocation runs at ~13 GB/s on tmpfs = ~265 4K frames/s whatever the GPU count (DESIGN.md)",
                "trace": pl[-1].split("pipeline: ", 1)[1] if pl else None,
                "phases": [ln.split("rcgpu trace:", 1)[1].strip() for ln in r.stderr.splitlines() if "rcgpu trace:" in ln and "pipeline:" not in ln],
                "what": f"BASELINE config 3 (config 2 + audio): process start to exit of rcgpu-ffmpeg: {n_frames} x {width}x{height} RGB16 DPX on tmpfs ({R} distinct, hard-linked){audio_note} -> FFV1 slices={slices} -> MKV on tmpfs; started after the device had been idle for {os.environ.get('RCGPU_BENCH_E2E_IDLE', '4')} s (the driver wipes the memory the previous leg freed)"}, ok is not False
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cfg1_leg(api, synth, device):
    """BASELINE config 1's shape on the device: 24 x 2048x1556 RGB 10-bit FilledA BE DPX payloads, host memory in, host memory out,
    through rcgpu_ffv1_encode_sequence_memory (what the ffmpeg subprocess does for such a package), slice count = the reference's."""
    import numpy as np
    import oracle_binding as ob
    w, h, n, pixfmt = 2048, 1556, 24, synth.PIX_RGB10_FILLEDA_BE
    payloads = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, 3, 10, "film", seed=i), pixfmt, True)
        payloads.append(np.frombuffer(pl, dtype=np.uint8).copy())
    slices = api.lib().rcgpu_reference_slices(w, h, 10, 1)
    nh, nv = api.slices_to_grid(slices)
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, nh, nv, 1, 1, 0, device, 0, 0, 1, 3)
    out_cap = int(payloads[0].nbytes * 1.3) + (1 << 20)
    outs = [np.empty(out_cap, dtype=np.uint8) for _ in range(n)]
    best = None
    for _ in range(2):                                   # the second call finds the runtime warm
        t0 = time.perf_counter()
        st, sizes = api.encode_sequence_memory(cfg, [a.ctypes.data for a in payloads], n, [a.ctypes.data for a in outs], out_cap, batch=n, device_first=device, device_count=1)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    ok = all(bytes(outs[i][:sizes[i]]) == ob.encode_payload(p, payloads[i].tobytes(), line_bytes) for i in (0, n - 1))
    return {"workload": f"config 1 shape: {n} x {w}x{h} RGB 10-bit FilledA BE, slices={slices} ({nh}x{nv}), host buffers in and out (rcgpu_ffv1_encode_sequence_memory)",
            "value": round(n / wall, 2), "unit": "frames/s", "call_seconds": round(wall, 3), "pipeline_seconds": round(st.seconds, 3), "prepare_seconds": round(st.prepare_seconds, 3),
            "first_packet_seconds": round(st.first_packet_seconds, 3), "packet_bytes_avg": int(sum(sizes) / n),
            "compression_ratio": round(sum(sizes) / n / payloads[0].nbytes, 4), "packets_equal_to_oracle": {"frames_checked": 2, "ok": bool(ok)},
            "note": "24 frames are one short batch: the number is start-up (encoder buffers, pinned slots) plus one batch's latency, not a rate"}, ok


def encode_leg(torch, api, synth, dev, device, label, w, h, pixfmt, slices, kind, F, steps=3, little_endian=False, seed=77):
    """One more encode configuration, device-resident and timed like the headline (run-on steps, a device-wide synchronisation at the end),
    with the dominant kernel's roofline from this run's own HIP events and a parity check on the device (the decoder, itself pinned by the
    oracle and the reference in tests/, rebuilds the first 8 payloads)."""
    line_bytes = w * 6
    nh, nv = api.slices_to_grid(slices)
    D = min(8, F)
    base = make_frames(torch, D, w, h, kind, seed, dev)                   # D distinct pictures (big endian as generated) ...
    if little_endian:
        base = base.view(D, -1, 2).flip(-1).reshape(D, -1).contiguous()   # ... byte-swapped: little endian, as TIFF stores them
    frames = base.repeat((F + D - 1) // D, 1)[:F].contiguous()
    del base
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=F, device=device)
    stride = (enc.max_packet + 255) & ~255
    d_packets = torch.empty(F * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    ptrs = [frames[i].data_ptr() for i in range(F)]
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        enc.encode_device(ptrs, d_packets.data_ptr(), stride, d_sizes.data_ptr(), stream)
    run_on = True
    try:
        enc.set_run_on(True)         # the steps run on from one to the next, as the headline's do
    except Exception:
        run_on = False
    step(); enc.join(stream); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    enc.join(stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = enc.kernel_times(); launches = enc.kernel_launches(); flags = enc.error_flags()
    sizes = d_sizes.cpu().tolist()
    dec = api.Ffv1Decoder(w, h, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=D, device=device)
    outs = torch.empty((D, line_bytes * h), dtype=torch.uint8, device=dev)
    dec.decode_device([d_packets.data_ptr() + i * stride for i in range(D)], sizes[:D], [outs[i].data_ptr() for i in range(D)], stream)
    ok = api.compare_device_batch([outs[i].data_ptr() for i in range(D)], ptrs[:D], [line_bytes * h] * D, stream) == [-1] * D
    dec.close()
    payload = line_bytes * h
    packet_avg = sum(sizes) / F
    # (in run-on mode k_model's HIP events span the time it trickles through beside the previous batch at its low priority, not its work)
    dom = max((k for k in kt if not (run_on and k == "k_model")), key=lambda k: kt[k])
    nl = max(1, launches.get(dom, 1))
    alg = F * (payload + packet_avg) / nl
    ach = alg / (kt[dom] / nl * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: the PMC passes were taken at 4K film content, 64 slices (profiles/traffic.json); state gathers, write-backs
    # and stream bytes are per sample, so a frame of another size moves its samples' share -- for film content only
    per4k, stale = traffic_of(dom, "ffv1_gpu.hip")
    own = None if stale else traffic_json()[0].get("configs", {}).get("%dx%d/%d" % (w, h, slices), {}).get(dom, {}).get("per_frame_bytes")      # PMC passes of this very configuration
    traffic = None if kind != "film" else int(own * F / nl) if own else int(per4k * (w * h) / (W4K * H4K) * F / nl) if per4k else None
    rec = {"workload": f"{label}: {w}x{h} RGB 16-bit {'LE' if little_endian else 'BE'}, slices={slices} ({nh}x{nv}), {F} frames per step resident in HBM, content={kind}" + (", steps issued in run-on mode" if run_on else ""),
           "value": round(F * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 2), "packet_bytes_avg": int(packet_avg),
           "compression_ratio": round(packet_avg / payload, 4), "device_error_flags": flags, "decodes_to_source_on_device": bool(ok),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
                        "traffic_stale": stale, "traffic_note": ("PMC passes of this configuration (profiles/traffic.json: configs)" if own else
                                                                 "scaled by samples per frame from the 4K film PMC passes (profiles/traffic.json); null for other content"),
                        "algorithmic_bytes_per_launch": int(alg), "launch_ms": round(kt[dom] / nl, 3), "launches_per_step": nl,
                        "kernel_ms_per_step": {k: round(v, 3) for k, v in kt.items() if v > 0}}}
    enc.close()
    del frames, d_packets, outs
    torch.cuda.empty_cache()
    return rec, bool(ok) and flags == 0


def cfg4_leg(torch, api, synth, dev, device, steps=3, F=80):
    """BASELINE config 4's shape on one device: 8192x4320 RGB 16-bit LE (the payload of a single-strip TIFF), the reference's 576 slices."""
    w, h = 8192, 4320
    return encode_leg(torch, api, synth, dev, device, "config 4 shape on one GPU (TIFF payload)", w, h, synth.PIX_RGB16_LE, api.lib().rcgpu_reference_slices(w, h, 16, 1), "film", F, steps, little_endian=True)


def encode576_leg(torch, api, synth, dev, device, width, height):
    """The encode half at the slice count the reference itself picks for 4K 16-bit (slice_x = slice_y = 24 -> `-slices 576`,
    Lib/Uncompressed/DPX/DPX.cpp:428-441): nine times the chains per frame, so 40 frames fill the device (the pipeline's own choice, by chains)."""
    return encode_leg(torch, api, synth, dev, device, "config 2 at the reference's own slice count", width, height, synth.PIX_RGB16_BE, api.lib().rcgpu_reference_slices(width, height, 16, 1), "film", 40, 24, seed=9)      # 24 steps: 1.5 s inside the clock


def flat_leg(torch, api, synth, dev, device, width, height, slices, F):
    """SURVEY.md 8d's second content class: "flat" (constant colour per frame, the best case) at the headline's geometry."""
    return encode_leg(torch, api, synth, dev, device, "config 2, content class (ii) flat", width, height, synth.PIX_RGB16_BE, slices, "flat", F, 3, seed=1000)


def check576_leg(args, torch, api, synth, dev, device, width, height):
    """The check half at the slice count the reference itself picks for 4K 16-bit (576): nine times the chains per frame, so a ninth of
    the frames in flight gives the decoder the same number of chains."""
    pixfmt = synth.PIX_RGB16_BE
    line_bytes = width * 6
    slices = api.lib().rcgpu_reference_slices(width, height, 16, 1)
    nh, nv = api.slices_to_grid(slices)
    F = 64
    frames = make_frames(torch, F, width, height, "film", 5, dev)
    enc = api.Ffv1Encoder(width, height, pixfmt, line_bytes, nh, nv, 1, 1, max_batch=F, device=device)
    stride = (enc.max_packet + 255) & ~255
    d_packets = torch.empty(F * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    ptrs = [frames[i].data_ptr() for i in range(F)]
    stream = torch.cuda.current_stream().cuda_stream
    enc.encode_device(ptrs, d_packets.data_ptr(), stride, d_sizes.data_ptr(), stream)
    torch.cuda.synchronize()
    flags = enc.error_flags()
    sizes = d_sizes.cpu().tolist(); record = enc.config_record(); enc.close()
    # the product-level figure at the slice count RAWcooked itself asks FFmpeg for: the real reference with the device decoder linked in, 1000 frames
    linked = None
    if "cpu" in {x for x in args.legs.split(",") if x}:
        linked = linked_check_record(api, synth, record, frames, d_packets, stride, sizes, width, height, pixfmt, nframes=1000, variants={"device_decoder"})
    rec, ok = check_leg(args, torch, api, record, frames, d_packets, sizes, ptrs, stride, stream, F, width, height, line_bytes, nh, nv, pixfmt, device,
                        steps=2, warmup=1, cpu=False, check_batch=512)
    del frames, d_packets
    torch.cuda.empty_cache()
    out = {k: rec[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline") if k in rec}
    if linked:
        out["linked_check"] = linked
    return out, ok and flags == 0


def bind_to_numa_node(node: int) -> None:
    """Experiment: keep this process (and every thread the library starts) on the cores of one NUMA node, so that first-touch places
    pinned and pageable buffers there."""
    cpus = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus |= set(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("RCGPU_BENCH_BATCH", "336")), help="frames in flight per GPU per step")
    ap.add_argument("--segments", type=int, default=0, help="hand-over windows per slice between k_resolve and k_rangecode (0 = library default)")
    ap.add_argument("--rc-span", type=int, default=0, help="range coder mapping: 0 = the library's choice (whole slices from 300 wavefronts of chains), 1 = whole slices, >= 8 = split coder with spans of that many pieces")
    ap.add_argument("--run-on", type=int, default=1, help="1: the timed steps in the encoder's run-on mode (batch k+1 started while batch k is coded); 0: one batch at a time")
    ap.add_argument("--check-batch", type=int, default=1600, help="check leg: frames decoded per step (>= --batch)")
    ap.add_argument("--kind", default="film", choices=["film", "flat", "noise"])
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--legs", default="host,e2e,check,cpu,configs",
                    help="comma list of the extra records: host (host_pipeline), e2e (config 3), check (config 5), cpu (cpu_baseline), configs (config 1 and 4 shapes, 576 slices, flat content), "
                         "long (the linked reference on a 5000-frame sequence, ~100 s: not in the default set, which has to fit the driver's run); '' = none")
    ap.add_argument("--host-frames", type=int, default=3840, help="host_pipeline: frames per GPU")
    ap.add_argument("--host-lanes", type=int, default=1, help="host_pipeline: encoder instances per GPU, batches staggered")
    ap.add_argument("--host-readers", type=int, default=0)
    ap.add_argument("--host-writers", type=int, default=0)
    ap.add_argument("--host-slots", type=int, default=0)
    ap.add_argument("--e2e-frames", type=int, default=1000, help="e2e: frames of the sequence (BASELINE config 2: 1000)")
    ap.add_argument("--long-frames", type=int, default=5000, help="long: frames of the long-sequence product leg (0 = off)")
    ap.add_argument("--time-budget", type=float, default=float(os.environ.get("RCGPU_BENCH_BUDGET", "420")),
                    help="seconds after which the legs that are not part of the contract's line (long, flat, encode_576_slices) are skipped and say so")
    ap.add_argument("--dma-noise", action="store_true", help="experiment: pinned H2D + D2H copies at full rate on two side streams during the timed steps")
    ap.add_argument("--context-model", default="ffmpeg", choices=["ffmpeg", "compact"],
                    help="level maps of the 5-input context model: FFmpeg's (5063 contexts, states in HBM) or compact (338 contexts, states in LDS)")
    ap.add_argument("--check-offsets", default="", help="--mode check: comma list of byte offsets (multiples of 256, <= 64 MiB) at which the decoder's state arrays are placed in turn; "
                                                         "k_dec_slices' time at each is recorded (does the time depend on the addresses?)")
    ap.add_argument("--check-profile-out", default="", help="--mode check: write the per-launch HIP-event timings to this file (profiles/r05_check_partitioned.json)")
    ap.add_argument("--detail-out", default=DETAIL_FILE, help="file (beside this script unless absolute) that receives everything measured; the stdout line is the compact one")
    ap.add_argument("--mode", default="encode", choices=["encode", "check"],
                    help="check: BASELINE config 5 alone -- device FFV1 decode + inverse transform + byte compare + MD5 of the encoder's packets")
    args = ap.parse_args()
    if args.check_offsets:
        os.environ["RCGPU_DEC_STATES_SLACK"] = "1"
    t_bench = time.perf_counter()
    marks = []

    def mark(name):
        marks.append((name, round(time.perf_counter() - t_bench, 1)))
    if os.environ.get("RCGPU_NUMA_NODE"):
        bind_to_numa_node(int(os.environ["RCGPU_NUMA_NODE"]))
    legs = {x for x in args.legs.split(",") if x}
    if args.no_cpu_baseline:
        legs.discard("cpu")

    alias_n = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        n_phys = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
        if 1 <= n_phys < args.gpus:
            # fewer GPUs than asked for (the 1-GPU boxes this repository is developed on): the headline runs on the one device and the
            # `single_process_sharding` record drives the product's lane-per-device path over N ALIASES of it (rcgpu_sequence_options::device_aliases)
            alias_n = args.gpus
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not alias_n:
        # asked for several GPUs without a launcher: start one rank per GPU ourselves, the way the driver does
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    from rawcooked_amd import api, synth
    import bench_dist as rdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = rdist.init(world, rank, local_rank, "nccl", dev)          # None at world == 1

    def barrier():
        rdist.barrier(dist)

    def reduce_max(x):
        return rdist.max_over_ranks(dist, x, dev)

    width, height, F = args.width, args.height, args.batch
    pixfmt = synth.PIX_RGB16_BE
    line_bytes = width * 6
    nh, nv = api.slices_to_grid(args.slices)
    frames = make_frames(torch, F, width, height, args.kind, rank, dev)
    ctx = 2 if args.context_model == "compact" else 1
    enc = api.Ffv1Encoder(width, height, pixfmt, line_bytes, nh, nv, 1, ctx, max_batch=F, device=local_rank, segments=args.segments, rc_span=args.rc_span)
    stride = (enc.max_packet + 255) & ~255
    d_packets = torch.empty(F * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    ptrs = [frames[i].data_ptr() for i in range(F)]
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        enc.encode_device(ptrs, d_packets.data_ptr(), stride, d_sizes.data_ptr(), stream)

    # The timed steps run one behind the other as a caller with a long sequence on the device would issue them: in run-on mode
    # (rcgpu_ffv1_set_run_on) batch k+1 is modelled while batch k is coded and the range coder's chain runs through.  Every step
    # still does all of its work inside the timed region (the region ends with a device-wide synchronisation); --run-on 0 times the
    # steps one at a time, the device idle between them, as the line did until round 4.
    run_on = bool(args.run_on) and args.mode != "check" and args.context_model != "compact"
    if args.mode == "check":
        step(); torch.cuda.synchronize()
        sizes = d_sizes.cpu().tolist(); record = enc.config_record(); enc.close()
        rec, ok = check_leg(args, torch, api, record, frames, d_packets, sizes, ptrs, stride, stream, F, width, height, line_bytes, nh, nv, pixfmt, local_rank,
                            args.steps, args.warmup, cpu="cpu" in legs)
        if args.check_profile_out:
            r_ = rec["roofline"]
            json.dump({"what": "bench.py --mode check: HIP-event time of every launch of the timed steps, in the configuration the line times (the hash on eight CUs of its own, "
                               "the decoder on the other 248: CU-masked streams, ffv1_check.hip partition_streams); decode = k_dec_split + k_dec_crc + k_dec_slices of one batch of "
                               "frames_per_step frames, md5 = k_md5 of the batch before (beside the decoder), compare = k_compare_batch",
                       "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "frames_per_step": rec["config"]["frames_per_step_per_gpu"], "slices": nh * nv,
                       "per_launch_ms": r_["per_launch_ms"], "last_launch_kernel_ms": r_["kernel_ms"], "hash_on_cus_of_its_own": r_["hash_on_cus_of_its_own"],
                       **({"k_dec_slices_ms_by_states_offset": r_["k_dec_slices_ms_by_states_offset"], "k_dec_slices_ms_by_allocation": r_.get("k_dec_slices_ms_by_allocation")}
                          if "k_dec_slices_ms_by_states_offset" in r_ else {}),
                       "all_frames_identical_to_source": rec["config"]["all_frames_identical_to_source"]}, open(args.check_profile_out, "w"), indent=1)
        print(json.dumps(rec))
        api.lib().rcgpu_release_device_streams()      # a CU-masked stream alive at exit takes rocprofv3's trace with it (profiles/r05_rocprof_cumask.txt)
        sys.exit(0 if ok else 2)

    noise = None
    if args.dma_noise:
        import threading as _th
        stop = _th.Event()
        hp_in = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True); hp_out = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
        dd_in = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dd_out = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        moved = [0]

        def pump():
            # knobs for finding out what a copy pattern costs beside the kernels (all through hipMemcpyAsync on one stream per direction)
            mode = os.environ.get("RCGPU_DMA_NOISE", "both")             # both | up | down
            mb = float(os.environ.get("RCGPU_DMA_NOISE_MB", "1024"))     # bytes per copy
            per_sync = int(os.environ.get("RCGPU_DMA_NOISE_SYNC", "20")) # copies per direction between two host synchronisations
            nbuf = int(os.environ.get("RCGPU_DMA_NOISE_BUFS", "0"))      # > 0: that many separate pinned buffers per direction (hipHostMalloc)
            flags = int(os.environ.get("RCGPU_DMA_NOISE_FLAGS", "1"), 0) # hipHostMalloc flags for them (1 = portable)
            spread = int(os.environ.get("RCGPU_DMA_NOISE_SPREAD_GB", "1"))   # device region the copies walk through
            ev_every = int(os.environ.get("RCGPU_DMA_NOISE_EVENTS", "0"))    # > 0: record an event every so many copies
            sz = int(mb * (1 << 20)) & ~4095
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            s_up, s_dn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            dreg_in = torch.empty(spread << 30, dtype=torch.uint8, device=dev); dreg_out = torch.empty(spread << 30, dtype=torch.uint8, device=dev)
            if nbuf > 0:
                def pinned(n):
                    q = ctypes.c_void_p(); assert hip.hipHostMalloc(ctypes.byref(q), ctypes.c_size_t(n), ctypes.c_uint(flags)) == 0; return q.value
                hin = [pinned(sz) for _ in range(nbuf)]; hout = [pinned(sz) for _ in range(nbuf)]
            else:
                nbuf = max(1, (1 << 30) // sz)
                hin = [hp_in.data_ptr() + k * sz for k in range(nbuf)]; hout = [hp_out.data_ptr() + k * sz for k in range(nbuf)]
            evs = []
            for _ in range(64):
                e_ = ctypes.c_void_p(); hip.hipEventCreateWithFlags(ctypes.byref(e_), ctypes.c_uint(2)); evs.append(e_)
            nd = max(1, (spread << 30) // sz)
            k = 0
            while not stop.is_set():
                for _ in range(per_sync):
                    if mode != "down":
                        assert hip.hipMemcpyAsync(dreg_in.data_ptr() + (k % nd) * sz, hin[k % nbuf], sz, 1, s_up.cuda_stream) == 0; moved[0] += sz
                    if mode != "up":
                        assert hip.hipMemcpyAsync(hout[k % nbuf], dreg_out.data_ptr() + (k % nd) * sz, sz, 2, s_dn.cuda_stream) == 0; moved[0] += sz
                    k += 1
                    if ev_every and k % ev_every == 0:
                        hip.hipEventRecord(evs[(k // ev_every) % 64], s_up.cuda_stream); hip.hipEventRecord(evs[(k // ev_every + 32) % 64], s_dn.cuda_stream)
                s_up.synchronize(); s_dn.synchronize()
        noise = _th.Thread(target=pump); noise.start(); t_noise = time.perf_counter()

    # ---- the headline: device-resident steps, timed as the driver's contract says (barrier + synchronize on both sides, max over ranks)
    mode_probe = None
    if run_on:
        try:
            enc.set_run_on(True)
        except Exception as ex:      # the second bank does not fit beside this batch: one batch at a time
            print("bench: run-on mode not available: %s" % ex, file=sys.stderr); run_on = False
    if run_on:
        # three steps in either mode before the clock starts: the timed steps use the faster one (run-on mode needs hardware queues of
        # its own for five streams; where the runtime deals them differently it must not cost the line 40 %)
        def probe(on):
            enc.set_run_on(on); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                step()
            enc.join(stream); torch.cuda.synchronize()
            return (time.perf_counter() - t) / 3
        probe(True)
        mode_probe = {"run_on_ms_per_step": round(probe(True) * 1e3, 1), "one_batch_at_a_time_ms_per_step": round(probe(False) * 1e3, 1)}
        run_on = mode_probe["run_on_ms_per_step"] <= mode_probe["one_batch_at_a_time_ms_per_step"]
        enc.set_run_on(run_on)
    # Per-step times for the line's `step_ms`: the device time from the end of one batch to the end of the next, from events the library
    # records behind each batch's last kernel on the stream it ran on (rcgpu_ffv1_batch_intervals).  Nothing of the bench's own stands
    # between the batches: round 6's first attempt joined every batch on a side stream to carry an event, and that stream's barrier
    # packets shared a hardware queue with k_resolve's stream -- the next batch's first segment then waited for the whole previous batch,
    # 22 ms per step (profiles/r06_trace_run_on.txt).  The first entry's interval begins at the end of the last warm-up batch, which is
    # before the clock starts: the entries add up to a little more than the timed region.
    dt = rdist.timed_steps(dist, dev, step, args.steps, args.warmup, torch.cuda.synchronize)
    if run_on:
        enc.join(stream); torch.cuda.synchronize()
    step_ms = enc.batch_intervals(min(args.steps, 63))          # (fewer entries than steps when no batch at all went before the first timed one)
    if noise is not None:
        stop.set(); noise.join()
        print("bench: dma noise moved %.0f GB during warm-up and timed steps = %.1f GB/s beside the kernels" % (moved[0] / 1e9, moved[0] / 1e9 / max(1e-9, time.perf_counter() - t_noise)), file=sys.stderr)
    mark("headline steps")
    kt = enc.kernel_times()          # HIP events of the last timed step, recorded on the launch streams
    try:
        flags = enc.error_flags()    # the device-pointer API only enqueues: this is where an overflow would show (raises)
    except Exception:
        if not os.environ.get("RCGPU_BENCH_TIMING_BUILD"):      # timing builds (tools/sweep_lds.sh) code wrong bytes on purpose
            raise
        flags = -1

    sizes = d_sizes.cpu().tolist()
    decisions, _ = enc.stats()
    total_frames = world * F * args.steps
    fps = total_frames / dt
    payload_bytes = line_bytes * height
    packet_avg = sum(sizes) / len(sizes)
    hbm_in_use = round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 1e9, 1)

    verified = verified_ref = None
    R = min(F, 32)
    if rank == 0 and not args.no_verify:
        # parity spot check outside the timed region: packet 0 of the last step decodes to the source through the oracle
        import oracle_binding as ob
        p = ob.Params(width, height, pixfmt, nh, nv, 1, ctx)
        pk = bytes(d_packets[:sizes[0]].cpu().numpy())
        src = bytes(frames[0].cpu().numpy())
        verified = ob.decode_payload(p, pk, line_bytes) == src
        if not verified:
            print("bench: GPU packet does not decode to the source payload", file=sys.stderr)
            sys.exit(2)
        # ... and the real reference (when its binary travelled with the repo) accepts an MKV of the first packets
        rb = reference_check_baseline(api, synth, enc.config_record(), frames, d_packets, stride, sizes, width, height, pixfmt, nframes=4)
        verified_ref = None if rb is None else rb["sample"].endswith("no issue detected")
        if verified_ref is False:
            print("bench: the reference rejected the GPU packets", file=sys.stderr)
            sys.exit(2)

    result = None
    if rank == 0:
        # (in run-on mode k_model's HIP events span the time it trickles through beside the previous batch at its low priority, not its work)
        dom = max((k for k in kt if not (run_on and k == "k_model")), key=lambda k: kt[k]) if kt else None
        launches = enc.kernel_launches()
        roof = None
        if dom:
            # the dominant kernel runs once per segment: algorithmic bytes and duration are both per launch
            nl = max(1, launches.get(dom, 1))
            alg_bytes_launch = F * (payload_bytes + packet_avg) / nl
            launch_ms = kt[dom] / nl
            achieved = alg_bytes_launch / (launch_ms * 1e-3) / 1e9
            tjd = traffic_json()[0]
            per_frame, traffic_stale = traffic_of(dom, "ffv1_gpu.hip")
            traffic = int(per_frame * F / nl) if per_frame else None      # PMC-measured HBM bytes per launch (scales with the batch)
            note = tjd.get("note")
            issue = None
            try:      # VALU wave-instructions per frame (SQ_INSTS_VALU, profiles/) against the issue rates tools/valu_peak measured on this chip
                v = tjd.get("valu", {})
                per_frame = v.get("wave_instr_per_frame_split" if enc.kernel_launches().get("k_rc_range") else "wave_instr_per_frame_whole")
                if per_frame:
                    rate = per_frame * fps / world
                    issue = {"valu_wave_instr_per_frame": per_frame, "frac_of_half_rate_peak": round(rate / (1024 * v["half_rate_per_simd"]), 4),
                             "frac_of_full_rate_peak": round(rate / (1024 * v["full_rate_per_simd"]), 4), "note": v.get("note")}
            except Exception:
                issue = None
            # the roofline that binds: random 32-byte records gathered and written back (one per sample), against what tools/gather_region
            # measured the chip to sustain in exactly that pattern; and the kernel's floor with that traffic served by the L2
            req = None
            try:
                rc_, fl_ = tjd.get("request_ceiling", {}), tjd.get("resolve_floor", {})
                samples = width * height * 3
                rate = samples * fps / world / 1e9
                req = {"records_G_per_s": round(rate, 2), "ceiling_G_per_s": rc_["G_records_per_s"], "uniform_address_ceiling_G_per_s": rc_.get("uniform_G_records_per_s"),
                       "request_frac": round(rate / rc_["G_records_per_s"], 4),
                       "floor_ms": round(fl_["ms_per_step_alone"] / fl_["launches_per_step"] * F / fl_["frames_per_step"], 3), "floor_what": fl_.get("what"), "ceiling_what": rc_.get("what")}
            except Exception:
                req = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_stale": traffic_stale,
                    "traffic_measured_on": tjd.get("measured_on"), "issue_frac": issue,
                    "request_frac": req["request_frac"] if req else None, "floor_ms": req["floor_ms"] if req else None, "requests": req,
                    "algorithmic_bytes_per_launch": int(alg_bytes_launch), "launch_ms": round(launch_ms, 3), "launches_per_step": nl,
                    "kernel_ms_per_step": {k: round(v, 3) for k, v in kt.items() if v > 0},
                    **({"kernel_ms_note": "run-on mode: the kernels of consecutive steps overlap (their times add up to more than a step), and k_model's figure is "
                                          "the time it takes to trickle through beside the previous batch on its low-priority stream; alone it takes 29 ms"} if run_on else {}),
                    "note": note or "entropy coding: serial per slice and per context, bound by instruction issue rather than bytes (DESIGN.md section 5)"}
        result = {
            "metric": "4K-DCI 16-bit DPX->FFV1 frames/sec", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "h2d_included": False,      # `value`: payloads resident in HBM when the clock starts; `kernel_metric` below is SURVEY.md 8d's H2D-inclusive figure
            "config": {"workload": f"BASELINE config 2: {width}x{height} RGB 16-bit BE DPX payload -> FFV1 v3 intra, slices={args.slices} ({nh}x{nv}), "
                                   f"coder=1 context=1 ({args.context_model} level maps) slicecrc=1, content={args.kind}; the other configs of BASELINE.json: "
                                   f"config 1 shape -> `configs.cfg1`, config 3 (config 2 + 6-ch WAV, files to MKV) -> `e2e`, config 4 shape on one GPU -> `configs.cfg4`, "
                                   f"config 5 (--check: decode + compare + MD5 on the device) -> `check` (64 slices) and `configs.check_576_slices`",
                       "range_coder": "split (k_rc_range + k_rangecode<true> + k_rc_tails)" if launches.get("k_rc_range") else "one lane per slice",
                       "frames_per_step_per_gpu": F, "parallelism": f"frame-sharded x{world}, no collective",
                       "steps_issued": "run-on: batch k+1 is modelled and started while batch k is coded (rcgpu_ffv1_set_run_on)" if run_on else "one batch at a time",
                       **({"steps_issued_probe": mode_probe} if mode_probe else {}),
                       "packet_bytes_avg": int(packet_avg), "compression_ratio": round(packet_avg / payload_bytes, 4),
                       "decisions_per_frame": int(decisions / F) if decisions else None, "verified_vs_oracle": verified, "verified_by_reference": verified_ref,
                       "device_error_flags": flags, "hbm_in_use_gb": hbm_in_use},
            "roofline": roof,
        }

    # ---- the extra legs.  What they need of the headline run is kept on the host; the encoder's 170 GB go back first.
    record = enc.config_record()
    host_ring = [frames[i].cpu().numpy() for i in range(R)] if (legs & {"host", "e2e", "cpu", "long"}) else []
    expect = [bytes(d_packets[i * stride:i * stride + sizes[i]].cpu().numpy()) for i in range(R)] if (legs & {"host", "e2e", "cpu"}) else []
    src0 = bytes(frames[0].cpu().numpy()) if "cpu" in legs and not host_ring else None
    cfg = api.Ffv1Config(width, height, pixfmt, line_bytes, nh, nv, 1, ctx, 0, local_rank, args.segments, 0, 1, 3)
    ok_all = True
    if "check" in legs and world == 1:
        keep_pk = d_packets
    else:
        keep_pk = None
    enc.close()
    if keep_pk is None:
        del d_packets
    torch.cuda.empty_cache()

    if "host" in legs:
        # the encoder above has just given some 200 GB back, which the driver wipes in the background (measured: an allocation behind a free
        # of 39 GB waits 3 s): the leg starts on an idle device, as the product legs below do
        time.sleep(max(10.0, float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4"))))
        if os.environ.get("RCGPU_DMA_NOISE_HOST"):       # experiment: the copy pump beside the host pipeline -- do the engines have room?
            stop.clear(); moved[0] = 0; t_noise = time.perf_counter()
            noise = threading.Thread(target=pump); noise.start()
        hp, ok = host_pipeline_leg(api, cfg, host_ring, args.host_frames, F, expect, barrier, reduce_max, args.host_lanes, args.host_readers, args.host_writers, args.host_slots)
        ok_all &= ok
        if os.environ.get("RCGPU_DMA_NOISE_HOST"):
            stop.set(); noise.join()
            print("bench: dma noise beside the host pipeline: %.0f GB = %.1f GB/s" % (moved[0] / 1e9, moved[0] / 1e9 / max(1e-9, time.perf_counter() - t_noise)), file=sys.stderr)
        n_loc, dt_loc, dt_all = hp.pop("_local")
        rows = rdist.gather_floats(dist, [n_loc, dt_loc, hp["h2d_GBps"], hp["d2h_GBps"], hp["first_packet_seconds"], hp["steady_frames_per_second"]], dev)
        if result is not None:
            # per rank, so that a multi-GPU run shows WHERE the GPUs contend: every payload and packet crosses host memory about four times
            # (reader copy in, DMA out, DMA in, writer copy out)
            hp["per_rank"] = [{"rank": r, "frames": int(v[0]), "seconds": round(v[1], 3), "frames_per_second": round(v[0] / v[1], 2), "h2d_GBps": v[2], "d2h_GBps": v[3],
                               "first_packet_seconds": v[4], "steady_frames_per_second": v[5]} for r, v in enumerate(rows)]
            hp["host_memory_GBps_all_ranks"] = round(sum(2 * (v[2] + v[3]) for v in rows), 1)
            hp["value"] = round(world * n_loc / dt_all, 2); hp["unit"] = "frames/s"; hp["n_gpus"] = world
            hp["fraction_of_device_resident"] = round(hp["value"] / fps, 3)
            result["host_pipeline"] = hp
            mark("host_pipeline (pageable)")
        # the same with the ring page-locked by the caller: SURVEY.md 8d's own wording
        time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
        hq, ok = host_pipeline_leg(api, cfg, host_ring, args.host_frames, F, expect, barrier, reduce_max, args.host_lanes, args.host_readers, args.host_writers, args.host_slots, pinned=True)
        ok_all &= ok
        n_q, _, dt_q = hq.pop("_local")
        if result is not None:
            hq["value"] = round(world * n_q / dt_q, 2); hq["unit"] = "frames/s"; hq["n_gpus"] = world
            hq["fraction_of_device_resident"] = round(hq["value"] / fps, 3)
            result["host_pipeline_pinned_inputs"] = hq
        # ... and on BASELINE config 2's own 1000 frames (three batches: mostly the pipeline's fill and drain)
        time.sleep(float(os.environ.get("RCGPU_BENCH_E2E_IDLE", "4")))
        h1, ok = host_pipeline_leg(api, cfg, host_ring, 1000, F, expect, barrier, reduce_max, args.host_lanes, args.host_readers, args.host_writers, args.host_slots, pinned=True)
        ok_all &= ok
        n_1, _, dt_1 = h1.pop("_local")
        if result is not None:
            h1["value"] = round(world * n_1 / dt_1, 2); h1["unit"] = "frames/s"; h1["n_gpus"] = world
            result["host_pipeline_pinned_inputs_1000_frames"] = h1
        if result is not None and "host_pipeline" in result:
            hp = result["host_pipeline"]
            result["kernel_metric"] = {"value": h1["value"], "unit": "frames/s", "h2d_included": True, "frames": 1000, "long_run_value": hq["value"], "long_run_frames": args.host_frames,
                                       "from_pageable_inputs": hp["value"],
                                       "what": "SURVEY.md 8d 'kernel-only fps (inputs resident in pinned host memory, H2D included)' on BASELINE config 2's own 1000 frames: payloads start in "
                                               "pinned host memory and packets end in pageable host memory, uploads / coding / downloads overlapped (host_pipeline_pinned_inputs_1000_frames; "
                                               "long_run_value: the same on --host-frames frames, host_pipeline_pinned_inputs); "
                                               "from_pageable_inputs: the payloads start in pageable memory and reader threads copy them into pinned slots first (host_pipeline)"}
    if (world > 1 or alias_n > 1) and "e2e" in legs:
        # ---- N JOBS SIDE BY SIDE, N Matroska files: the way a node pays off under the single-file ceiling (one job writes one file at ~13 GB/s whatever
        # the GPU count), and the reference's own way to use one (GNU parallel, Doc/Case_study.md:81): every rank -- or, on a one-GPU box, every alias --
        # runs its own `rcgpu-ffmpeg` on its own device, files to MKV, all started together; every block of every file against the N = 1 run's packets.
        nj = alias_n if alias_n > 1 else world
        n_job = max(64, args.e2e_frames // (2 if alias_n > 1 else 1))
        barrier()
        t0 = time.perf_counter()
        if alias_n > 1:          # one GPU: the jobs share it (device 0) and its memory -- each takes a batch that fits beside the others
            import threading as _th
            recs = [None] * nj

            def one(j):
                recs[j] = ({"error": "did not run"}, False)
                recs[j] = e2e_leg(synth, host_ring, width, height, n_job, args.slices, expect, with_audio=False, device=0, tag="_job%d" % j, batch=max(8, F // (2 * nj)))
            ths = [_th.Thread(target=one, args=(j,)) for j in range(nj)]
            [t.start() for t in ths]; [t.join() for t in ths]
            rows = [[j, r_[0].get("frames", 0), r_[0].get("seconds", 0.0), 1.0 if r_[0].get("all_blocks_identical_to_device_resident_run") else 0.0] for j, r_ in enumerate(recs)]
            errs = [r_[0].get("error") or r_[0].get("skipped") for r_ in recs]
        else:
            try:         # whatever goes wrong in one rank's job is a row of zeros and an error text, not a rank that leaves the others waiting at the gather
                rec_, ok_ = e2e_leg(synth, host_ring, width, height, n_job, args.slices, expect, with_audio=False, device=local_rank, tag="_rank%d" % rank)
            except Exception as e:
                rec_ = {"error": str(e)[-300:]}
            rows = rdist.gather_floats(dist, [rank, rec_.get("frames", 0), rec_.get("seconds", 0.0), 1.0 if rec_.get("all_blocks_identical_to_device_resident_run") else 0.0], dev)
            errs = [rec_.get("error") or rec_.get("skipped")]
        wall = reduce_max(time.perf_counter() - t0)
        if result is not None:
            jobs = [{"job": int(v[0]), "device": 0 if alias_n > 1 else int(v[0]), "frames": int(v[1]), "seconds": round(v[2], 3), "frames_per_second": round(v[1] / v[2], 2) if v[2] else None,
                     "all_blocks_identical_to_the_n1_runs_packets": bool(v[3])} for v in rows]
            slowest = max((j["seconds"] for j in jobs), default=0.0)
            ok_jobs = all(j["all_blocks_identical_to_the_n1_runs_packets"] for j in jobs) and len(jobs) == nj
            ok_all &= ok_jobs
            result["jobs_side_by_side"] = {
                "value": round(sum(j["frames"] for j in jobs) / slowest, 2) if slowest else None, "unit": "frames/s", "jobs": nj, "frames_per_job": n_job, "slowest_job_seconds": slowest,
                "wall_seconds_with_setup": round(wall, 2), "per_job": jobs, "errors": [e for e in errs if e] or None,
                "what": (f"{nj} rcgpu-ffmpeg processes started together, each {n_job} x {width}x{height} RGB16 DPX on tmpfs -> FFV1 slices={args.slices} -> its own MKV on tmpfs, "
                         + ("all on the ONE GPU of this box (RCGPU_DEVICES=0,1; they share its memory and its time: this checks the path, the number is not a scaling figure)" if alias_n > 1
                            else "job r on GPU r (RCGPU_DEVICES=r,1)") + "; value = all frames / the slowest job's process start to exit")}
        barrier()
    if (world > 1 or alias_n > 1) and "host" in legs:
        # ---- the PRODUCT's sharding: one process, one sequence, a lane per device (frame i -> GPU by batch, one placer in frame
        # order; rawcooked_amd/csrc/pipeline.hip), next to the per-rank rows above, which are N independent jobs (the reference's own way to
        # use a node: GNU parallel, Doc/Case_study.md:81).  Rank 0 drives all N devices while the other ranks wait at the barrier.
        barrier()
        if rank == 0:
            try:      # a report beside the contract's line: whatever goes wrong here (a launcher that hides the other devices from rank 0, say) is shown, not fatal
                ndv = alias_n if alias_n > 1 else world
                n_seq = (args.host_frames // 2) * (1 if alias_n > 1 else world)
                sp, ok = host_pipeline_leg(api, cfg, host_ring, n_seq, (F // ndv) if alias_n > 1 else F, expect, lambda: None, lambda x: x, 1, args.host_readers, args.host_writers, args.host_slots,
                                           device_first=0, device_count=ndv, aliases=alias_n)
                ok_all &= ok
                n_loc, dt_loc, _ = sp.pop("_local")
                sp["value"] = round(n_loc / dt_loc, 2); sp["unit"] = "frames/s"; sp["devices"] = ndv
                # every lane's pinned download ring on the NUMA node its device hangs on (-1 = the kernel did not say)
                sp["pinned_rings_on_their_devices_nodes"] = all(l["pinned_ring_on_node"] == l["numa_node"] or l["numa_node"] < 0 or l["pinned_ring_on_node"] < 0 for l in sp.get("lanes", []))
                sp["lanes_per_device_seen"] = sorted({l["device"] for l in sp.get("lanes", [])})
                sp["what"] = (f"ONE process, ONE sequence of {n_seq} frames, a lane per device over {ndv} " + ("aliases of the one GPU of this box (the lanes share its memory and its kernels' time: "
                              "this checks the path, the number is not a scaling figure)" if alias_n > 1 else "GPUs") + ": batches dealt to the lanes in turn, no collective, packets placed in frame order "
                              "and compared with the N = 1 run's; lanes grouped by their device's NUMA node")
            except Exception as e:
                sp = {"error": str(e)[-300:]}
            if result is not None:
                result["single_process_sharding"] = sp
        barrier()
    if world > 1:
        # the process group is RCCL (torch's "nccl" backend on ROCm) and every rank is in it: an all-reduce of ones over the ranks' devices
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        if result is not None:
            result["rccl"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_counted_by_all_reduce": int(ones.item()),
                              "saw_every_rank": int(ones.item()) == world == dist.get_world_size(), "used_for": "barriers and the max-over-ranks of the timings only: the encode path has no collective"}
            ok_all &= result["rccl"]["saw_every_rank"]
    if rank == 0 and world > 1 and result is not None:
        result["single_job_note"] = ("frames shard over the GPUs with no collective; a single JOB, though, writes ONE Matroska file, whose page allocation runs at "
                                     "~13 GB/s = ~265 4K frames/s whatever the GPU count (e2e record at N = 1): N GPUs pay off for N jobs side by side or for a caller that keeps packets in memory")
    if rank == 0 and world == 1:
        if "e2e" in legs:
            rec, ok = e2e_leg(synth, host_ring, width, height, args.e2e_frames, args.slices, expect)
            ok_all &= ok
            result["e2e"] = rec
            mark("e2e")
        if "check" in legs:
            # (The device-resident part before the product-level records.  Running it FIRST among the extra legs was tried in round 5 -- k_dec_slices' time
            # depends on where its allocation lands, profiles/r05_check_allocations.json -- and undone: the host pipeline that then follows the decoder's 240 GB
            # free runs at 365 instead of 600 frames/s, its transfers at 19 instead of 35 GB/s, whatever the pause between them.)
            rec, ok = check_leg(args, torch, api, record, frames, keep_pk, sizes, ptrs, stride, stream, F, width, height, line_bytes, nh, nv, pixfmt, local_rank,
                                steps=2, warmup=1, cpu=False)
            ok_all &= ok
            if "cpu" in legs:
                only = set(filter(None, os.environ.get("RCGPU_LINKED_VARIANTS", "").split(","))) or None      # (measuring: some of the variants only)
                lr = linked_check_record(api, synth, record, frames, keep_pk, stride, sizes, width, height, pixfmt, variants=only)
                cr = reference_check_baseline(api, synth, record, frames, keep_pk, stride, sizes, width, height, pixfmt, parallel=max(1, min(8, usable_cores() // 32)))
                if lr:
                    rec["linked_check"] = lr
                if cr:
                    rec["cpu_baseline"] = cr
            result["check"] = {k: rec[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "cpu_baseline", "linked_check") if k in rec}
            mark("check + linked_check + reference check")
            if result["check"].get("config", {}).get("all_frames_identical_to_source"):
                result["config"]["packets_verified"] = (f"all {F} packets of the last timed step decode to their sources on the device (check record: byte compare + MD5); "
                                                        "the first ones also through the oracle and the reference binary (verified_vs_oracle, verified_by_reference), "
                                                        "and the CPU baseline's oracle packets are compared with the device's by MD5 (cpu_baseline.gpu_packets_equal_to_oracle)")
        del frames, keep_pk
        torch.cuda.empty_cache()
        if "configs" in legs:
            cfgs = {}
            for name, fn in (("cfg1", lambda: cfg1_leg(api, synth, local_rank)), ("cfg4", lambda: cfg4_leg(torch, api, synth, dev, local_rank)),
                             ("encode_576_slices", lambda: encode576_leg(torch, api, synth, dev, local_rank, width, height)),
                             ("flat", lambda: flat_leg(torch, api, synth, dev, local_rank, width, height, args.slices, F)),
                             ("check_576_slices", lambda: check576_leg(args, torch, api, synth, dev, local_rank, width, height))):
                if name in ("encode_576_slices", "flat") and time.perf_counter() - t_bench > args.time_budget:
                    cfgs[name] = {"skipped": "time budget of %.0f s used up (--time-budget)" % args.time_budget}
                    continue
                try:
                    cfgs[name], ok = fn()
                    ok_all &= ok
                except Exception as e:          # a sub-record is a report: its failure is shown, the headline stays
                    cfgs[name] = {"error": str(e)[-300:]}
                    ok_all = False
            result["configs"] = cfgs
            mark("configs")
        if "cpu" in legs:
            import hashlib
            cb, ok = cpu_baseline([a.tobytes() for a in host_ring] if host_ring else [src0], line_bytes, width, height, [hashlib.md5(e).hexdigest() for e in expect] if expect else [])
            ok_all &= ok
            result["cpu_baseline"] = cb
            mark("cpu_baseline")
    if rank == 0 and world == 1 and "long" in legs and args.long_frames > 0 and host_ring:
        if time.perf_counter() - t_bench > args.time_budget:
            result["long_sequence"] = {"skipped": "time budget of %.0f s used up (--time-budget)" % args.time_budget}
        else:
            result["long_sequence"], ok = long_sequence_leg(synth, host_ring, width, height, args.long_frames, pixfmt)
            ok_all &= ok
            mark("long_sequence")
    if rank == 0 and result is not None:
        # The contract's numbers where the driver keeps them: flat numeric keys inside `config` (the records they come from follow in the line)
        def num(*path):
            d = result
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d if isinstance(d, (int, float)) and not isinstance(d, bool) else None
        c = result["config"]
        c["h2d_inclusive_fps"] = num("kernel_metric", "long_run_value")                          # inputs in pinned host memory, H2D included; 3840 frames
        c["h2d_inclusive_fps_frames"] = args.host_frames if c["h2d_inclusive_fps"] is not None else None
        c["h2d_inclusive_fps_from_pageable"] = num("host_pipeline", "value")
        c["h2d_inclusive_fps_1000_frames"] = num("host_pipeline_pinned_inputs_1000_frames", "value")   # ... on config 2's own 1000 frames
        c["h2d_inclusive_steady_fps"] = num("host_pipeline_pinned_inputs", "steady_frames_per_second")
        c["e2e_fps"] = num("e2e", "value")                                                        # config 3: files + WAV -> MKV, process start to exit
        c["check_fps"] = num("check", "value")                                                    # config 5, device-resident, 64 slices
        c["check_roofline_frac"] = num("check", "roofline", "frac")
        c["check_request_frac"] = num("check", "roofline", "request_frac")
        c["check_576_slices_fps"] = num("configs", "check_576_slices", "value")
        c["linked_check_fps"] = num("check", "linked_check", "device_decoder", "value")           # the real reference + the device decoder, 1000 frames
        c["linked_check_576_slices_fps"] = num("configs", "check_576_slices", "linked_check", "device_decoder", "value")
        c["linked_check_cpu_pool_fps"] = num("check", "linked_check", "reference_cpu_pool", "value")
        c["whole_product_fps"] = num("check", "linked_check", "whole_product", "value")
        c["encode_576_slices_fps"] = num("configs", "encode_576_slices", "value")                 # config 2 at the reference's own slice count
        c["encode_576_slices_roofline_frac"] = num("configs", "encode_576_slices", "roofline", "frac")
        c["flat_fps"] = num("configs", "flat", "value")                                           # SURVEY 8d content class (ii)
        c["flat_roofline_frac"] = num("configs", "flat", "roofline", "frac")
        c["cfg4_8k_fps"] = num("configs", "cfg4", "value")
        c["cfg4_8k_roofline_frac"] = num("configs", "cfg4", "roofline", "frac")
        c["long_frames"] = num("long_sequence", "frames")
        c["long_encode_product_fps"] = num("long_sequence", "analysis_and_encode", "value")
        c["long_encode_steady_fps"] = num("long_sequence", "analysis_and_encode", "encode_steady_frames_per_second")
        c["long_linked_check_fps"] = num("long_sequence", "linked_check", "value")
        c["long_linked_check_steady_fps"] = num("long_sequence", "linked_check", "steady_frames_per_second")
        c["jobs_side_by_side_fps"] = num("jobs_side_by_side", "value")
        c["single_process_sharding_fps"] = num("single_process_sharding", "value")
        c["rccl_ranks"] = num("rccl", "ranks_counted_by_all_reduce")
        c["cpu_baseline_fps"] = num("cpu_baseline", "value")
        c["bench_wall_seconds"] = round(time.perf_counter() - t_bench, 1)
        result["legs_done_at_seconds"] = {k: v for k, v in marks}
    if rank == 0:
        # The driver parses the LAST line of stdout and keeps 8 KB of it: one compact object there; everything measured goes to
        # bench_detail.json beside this script (and gpurun_out/, which travels back from the GPU box) and to stderr.
        result["step_ms"] = [round(x, 2) for x in step_ms]
        detail = json.dumps(result, indent=1)
        outs = [args.detail_out if os.path.isabs(args.detail_out) else os.path.join(ROOT, args.detail_out)]
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            outs.append(os.path.join(ROOT, "gpurun_out", os.path.basename(args.detail_out)))
        for path in outs:
            try:
                with open(path, "w") as fh:
                    fh.write(detail + "\n")
            except OSError as ex:
                print("bench: cannot write %s: %s" % (path, ex), file=sys.stderr)
        # (stderr stays short as well: the driver's record joins the tails of both streams)
        print("bench: %s %s %s, %s ms per step; detail record: %d bytes in %s" % (result["metric"], result["value"], result["unit"], result["ms_per_step"], len(detail), ", ".join(outs)),
              file=sys.stderr)
        sys.stderr.flush()
        print(compact_line(result, step_ms, os.path.basename(args.detail_out)))
        sys.stdout.flush()
    api.lib().rcgpu_release_device_streams()          # (see --mode check above)
    rdist.finish(dist)
    if not ok_all:
        sys.exit(2)


if __name__ == "__main__":
    main()
