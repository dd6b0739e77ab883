"""The upload / encode / download pipeline (rcgpu_ffv1_encode_sequence, rawcooked_amd/csrc/pipeline.hip): what the ffmpeg process
started at Source/CLI/Output.cpp:356 does with one picture sequence.  Packets must equal the oracle's, whatever the batching."""
import ctypes as C
import threading

import pytest

from rawcooked_amd import api, synth
import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _sequence(w, h, pixfmt, n, kind="film"):
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    out = []
    for i in range(n):
        pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, kind, seed=100 + i), pixfmt, True)
        out.append(pl)
    return out, line_bytes


@pytest.mark.parametrize("n,batch,in_ring,out_ring", [(37, 8, 3, 1 << 20), (5, 16, 0, 0), (64, 0, 0, 0), (1, 0, 0, 0)])
def test_sequence_equals_oracle(n, batch, in_ring, out_ring):
    w, h, pixfmt = 192, 96, synth.PIX_RGB16_BE
    payloads, line_bytes = _sequence(w, h, pixfmt, n)
    keep = [C.create_string_buffer(p, len(p)) for p in payloads]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 3, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    got, lock, order = {}, threading.Lock(), []

    def read_frame(frame, dst, nbytes):
        assert nbytes == len(payloads[frame])
        C.memmove(dst, keep[frame], nbytes)
        return 0

    def packet_done(frame, data, size):
        b = C.string_at(data, size)
        with lock:
            got[frame] = b
            order.append(frame)
        return 0

    st, rec = api.encode_sequence(cfg, n, read_frame, packet_done, batch=batch, in_ring_frames=in_ring, out_ring_bytes=out_ring, readers=3, writers=2)
    assert st.frames == n and len(got) == n
    assert st.packet_bytes == sum(len(v) for v in got.values())
    p = ob.Params(w, h, pixfmt, 3, 2, 1, 1)
    for i in range(n):
        assert got[i] == ob.encode_payload(p, payloads[i], line_bytes), f"packet {i} differs from the oracle's"
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 3, 2, 1, 1, max_batch=1)
    assert rec == enc.config_record()
    enc.close()


def test_place_packet_is_called_in_frame_order_and_filled():
    w, h, pixfmt, n = 128, 64, synth.PIX_RGB10_FILLEDA_BE, 23
    payloads, line_bytes = _sequence(w, h, pixfmt, n)
    keep = [C.create_string_buffer(p, len(p)) for p in payloads]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 2, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    arena = C.create_string_buffer(n * (1 << 16))
    base = C.addressof(arena)
    placed, sizes = [], {}

    def place(frame, size):
        placed.append(frame)
        sizes[frame] = size
        return base + frame * (1 << 16)

    def done(frame, data, size):
        assert data == base + frame * (1 << 16) and size == sizes[frame]
        return 0

    api.encode_sequence(cfg, n, lambda f, d, nb: C.memmove(d, keep[f], nb) and 0, done, place_packet=place, batch=5)
    assert placed == list(range(n))
    p = ob.Params(w, h, pixfmt, 2, 2, 1, 1)
    for i in range(n):
        assert arena.raw[i * (1 << 16):i * (1 << 16) + sizes[i]] == ob.encode_payload(p, payloads[i], line_bytes)


def test_a_failing_reader_ends_the_job_with_its_code():
    w, h, pixfmt, n = 64, 48, synth.PIX_RGB16_BE, 40
    payloads, line_bytes = _sequence(w, h, pixfmt, 1)
    keep = C.create_string_buffer(payloads[0], len(payloads[0]))
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 2, 2, 1, 1, 0, 0, 0, 0, 1, 3)

    def read_frame(frame, dst, nbytes):
        if frame == 17:
            return 77
        C.memmove(dst, keep, nbytes)
        return 0

    with pytest.raises(api.RcgpuError):
        api.encode_sequence(cfg, n, read_frame, lambda f, d, s: 0, batch=4)
    # the library is usable afterwards
    st, _ = api.encode_sequence(cfg, 3, lambda f, d, nb: C.memmove(d, keep, nb) and 0, lambda f, d, s: 0, batch=4)
    assert st.frames == 3


def test_device_pointer_api_reports_a_slice_that_outgrew_its_buffer():
    """rcgpu_ffv1_encode_device only enqueues work; rcgpu_ffv1_last_error_flags is how its callers learn about an overflow.  Forced
    here with RCGPU_TEST_CBUF_DIV (slice buffers a fraction of their normal size) and incompressible content."""
    import os
    import torch
    w, h, pixfmt = 256, 128, synth.PIX_RGB16_BE
    bits, nc, _, _ = synth.PIX_INFO[pixfmt]
    pl, line_bytes = synth.pack_payload(synth.components(w, h, nc, bits, "noise", seed=5), pixfmt, True)
    enc = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 2, 2, 1, 1, max_batch=1, slice_buffer_div=64)
    d = torch.frombuffer(bytearray(pl), dtype=torch.uint8).cuda()
    pk = torch.empty(enc.max_packet, dtype=torch.uint8, device="cuda")
    sz = torch.zeros(1, dtype=torch.int64, device="cuda")
    enc.encode_device([d.data_ptr()], pk.data_ptr(), enc.max_packet, sz.data_ptr(), torch.cuda.current_stream().cuda_stream)
    with pytest.raises(api.RcgpuError, match="outgrew|does not fit"):
        enc.error_flags()
    enc.close()
    ok = api.Ffv1Encoder(w, h, pixfmt, line_bytes, 2, 2, 1, 1, max_batch=1)
    ok.encode_device([d.data_ptr()], pk.data_ptr(), ok.max_packet, sz.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert ok.error_flags() == 0
    ok.close()


def test_sequence_memory_host_buffers_in_and_out():
    """rcgpu_ffv1_encode_sequence_memory: host buffers on both ends, rings of buffers re-used (frame i = in[i % n_in], packet i ->
    out[i % n_out]); several batches, so the deferred gather and the batch k+1 / download k order are exercised."""
    import numpy as np
    w, h, pixfmt, n_in, n = 160, 90, synth.PIX_RGB16_BE, 5, 43
    payloads, line_bytes = _sequence(w, h, pixfmt, n_in)
    src = [np.frombuffer(p, dtype=np.uint8).copy() for p in payloads]
    out_cap = len(payloads[0]) * 2
    outs = [np.zeros(out_cap, dtype=np.uint8) for _ in range(n)]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 3, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    st, sizes = api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], n, [a.ctypes.data for a in outs], out_cap, batch=6)
    assert st.frames == n and st.batches == 8 and st.packet_bytes == sum(sizes)
    p = ob.Params(w, h, pixfmt, 3, 2, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert bytes(outs[i][:sizes[i]]) == want[i % n_in], f"packet {i}"
    # buffers too small for a packet: an error, not a silent truncation
    with pytest.raises(api.RcgpuError, match="does not fit"):
        api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], 4, [a.ctypes.data for a in outs], 64, batch=2)


@pytest.mark.parametrize("lanes,aliases,run_on", [(1, 0, 0), (2, 0, 0), (1, 2, 0), (1, 0, 1), (2, 0, 1), (1, 2, 1)])
def test_sequence_memory_from_pinned_frames(lanes, aliases, run_on):
    """rcgpu_sequence_options::frames_pinned (locate_frame in place of read_frame): the caller's frames are page-locked and the pipeline
    uploads from them -- no upload slots, no reader threads ("inputs resident in pinned host memory, H2D included").
    Same packets as from pageable frames, with one lane, two lanes on a device and a lane on each of two (aliased) devices -- and with the
    encoders in run-on mode (rcgpu_sequence_options::run_on: batch k+1 modelled behind its uploads while batch k is coded)."""
    import numpy as np
    import torch
    w, h, pixfmt, n_in, n = 160, 90, synth.PIX_RGB16_BE, 7, 45
    payloads, line_bytes = _sequence(w, h, pixfmt, n_in)
    pins = [torch.frombuffer(bytearray(p), dtype=torch.uint8).pin_memory() for p in payloads]
    out_cap = len(payloads[0]) * 2
    outs = [np.zeros(out_cap, dtype=np.uint8) for _ in range(n)]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 3, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    st, sizes = api.encode_sequence_memory(cfg, [t.data_ptr() for t in pins], n, [a.ctypes.data for a in outs], out_cap, batch=6, frames_pinned=1,
                                           lanes_per_device=lanes, device_aliases=aliases, device_count=2 if aliases else 0, run_on=run_on)
    assert st.frames == n and st.readers == 0 and st.packet_bytes == sum(sizes)
    p = ob.Params(w, h, pixfmt, 3, 2, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert bytes(outs[i][:sizes[i]]) == want[i % n_in], f"packet {i}"


@pytest.mark.parametrize("lanes,n,batch", [(2, 41, 6), (3, 50, 4), (4, 9, 8)])
def test_several_lanes_return_packets_in_frame_order(lanes, n, batch):
    """Several encoder instances ("lanes") whose batches run staggered -- on one device here, one per device on a multi-GPU node: batch b
    goes to lane b mod lanes, every lane has its own copy streams and download ring, and one placer hands the packets on in frame order."""
    import numpy as np
    w, h, pixfmt, n_in = 128, 72, synth.PIX_RGB16_BE, 7
    payloads, line_bytes = _sequence(w, h, pixfmt, n_in)
    src = [np.frombuffer(p, dtype=np.uint8).copy() for p in payloads]
    out_cap = len(payloads[0]) * 2
    outs = [np.zeros(out_cap, dtype=np.uint8) for _ in range(n)]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 2, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    st, sizes = api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], n, [a.ctypes.data for a in outs], out_cap, batch=batch, lanes_per_device=lanes)
    assert st.frames == n and st.devices == min(lanes, (n + 7) // 8)          # no more lanes than there is work for (8 frames each at least)
    p = ob.Params(w, h, pixfmt, 2, 2, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert bytes(outs[i][:sizes[i]]) == want[i % n_in], f"packet {i}"


def _numa_node_of_device(device):
    """What /sys says about the GPU's PCI function (the library reads the same file)."""
    import torch
    p = torch.cuda.get_device_properties(device)
    path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    try:
        node = int(open(path).read())
    except OSError:
        return -1
    nodes = 0
    while __import__("os").path.exists(f"/sys/devices/system/node/node{nodes}/cpulist"):
        nodes += 1
    return node if 0 <= node < nodes else -1


@pytest.mark.parametrize("aliases,n,batch", [(4, 70, 5), (2, 33, 6)])
def test_lanes_grouped_by_numa_node_keep_the_packets(aliases, n, batch):
    """numa = 2 (test hook): lane i is treated as attached to host node i mod nodes, so that the per-node pools -- pinned upload slots, reader
    and writer threads, frame cursors -- run side by side on a one-GPU box: same packets, in order, and every lane reports its node."""
    import numpy as np
    w, h, pixfmt, n_in = 128, 72, synth.PIX_RGB16_BE, 7
    payloads, line_bytes = _sequence(w, h, pixfmt, n_in)
    src = [np.frombuffer(p, dtype=np.uint8).copy() for p in payloads]
    out_cap = len(payloads[0]) * 2
    outs = [np.zeros(out_cap, dtype=np.uint8) for _ in range(n)]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 2, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    st, sizes = api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], n, [a.ctypes.data for a in outs], out_cap, batch=batch, device_count=aliases, device_aliases=aliases, numa=2,
                                           readers=5, writers=3, in_ring_frames=6)
    nodes = 0
    while __import__("os").path.exists(f"/sys/devices/system/node/node{nodes}/cpulist"):
        nodes += 1
    lanes = st.devices
    assert lanes == min(aliases, (n + 7) // 8)
    if nodes:
        assert list(st.lane_numa_node[:lanes]) == [i % nodes for i in range(lanes)] and st.host_groups == min(nodes, lanes)
    else:
        assert st.host_groups == 1
    p = ob.Params(w, h, pixfmt, 2, 2, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert bytes(outs[i][:sizes[i]]) == want[i % n_in], f"packet {i}"
    st1, sizes1 = api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], n, [a.ctypes.data for a in outs], out_cap, batch=batch, device_count=aliases, device_aliases=aliases, numa=1)
    assert st1.host_groups == 1 and list(st1.lane_numa_node[:lanes]) == [-1] * lanes and sizes1 == sizes


@pytest.mark.parametrize("aliases,first,count,n,batch", [(2, 0, 2, 41, 6), (4, 1, 3, 50, 4), (3, 0, 0, 37, 5)])
def test_a_lane_per_device_through_device_first_and_count(monkeypatch, aliases, first, count, n, batch):
    """The multi-GPU path proper -- a lane per entry of device_first / device_count, each with its own encoder, pinned ring, copy streams and
    events, one placer across them -- on a box with one GPU: rcgpu_sequence_options::device_aliases presents the device several times (the
    lanes share its memory, hence the explicit batch).  No 8-GPU node has run this code yet; this is the closest one GPU gets."""
    import numpy as np
    w, h, pixfmt, n_in = 128, 72, synth.PIX_RGB16_BE, 7
    payloads, line_bytes = _sequence(w, h, pixfmt, n_in)
    src = [np.frombuffer(p, dtype=np.uint8).copy() for p in payloads]
    out_cap = len(payloads[0]) * 2
    outs = [np.zeros(out_cap, dtype=np.uint8) for _ in range(n)]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 2, 2, 1, 1, 0, 0, 0, 0, 1, 3)
    st, sizes = api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], n, [a.ctypes.data for a in outs], out_cap, batch=batch, device_first=first, device_count=count, device_aliases=aliases)
    usable = (aliases - first) if count == 0 else min(count, aliases - first)
    assert st.frames == n and st.devices == min(usable, (n + 7) // 8), (st.devices, usable)
    # every lane sits on the NUMA node its device hangs on: one group here (aliases of one GPU), its pinned ring on that node
    node = _numa_node_of_device(0)
    assert st.host_groups == 1 and list(st.lane_device[:st.devices]) == [0] * st.devices and list(st.lane_numa_node[:st.devices]) == [node] * st.devices
    if node >= 0:
        assert all(pn in (-1, node) for pn in st.lane_pinned_node[:st.devices]), (node, list(st.lane_pinned_node[:st.devices]))
    p = ob.Params(w, h, pixfmt, 2, 2, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert bytes(outs[i][:sizes[i]]) == want[i % n_in], f"packet {i}"
    with pytest.raises(api.RcgpuError, match="outside"):
        api.encode_sequence_memory(cfg, [a.ctypes.data for a in src], 4, [a.ctypes.data for a in outs], out_cap, batch=2, device_first=aliases, device_count=1, device_aliases=aliases)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RCGPU_SOAK_PIPE", "6"))))      # soak: RCGPU_SOAK_PIPE=200
def test_random_pipeline_shapes_equal_the_oracle(seed, monkeypatch):
    """Sequence lengths, batch sizes, slot and ring sizes, thread counts, lanes and copy streams drawn at random: the packets are the
    oracle's whatever the shape (ordering of groups over several copy streams, slot recycling, ring wrap-around, evened batches)."""
    import random
    r = random.Random(1000 + seed)
    w, h = r.choice([(64, 48), (96, 64), (160, 80)])
    pixfmt = r.choice([synth.PIX_RGB16_BE, synth.PIX_RGB10_FILLEDA_BE, synth.PIX_RGB8])
    n = r.choice([1, 2, 7, 19, 33, 50, 77])
    nh, nv = r.choice([(1, 1), (2, 2), (3, 2)])
    copy_streams = r.choice([1, 2, 3])
    payloads, line_bytes = _sequence(w, h, pixfmt, min(n, 12), kind=r.choice(["film", "noise", "flat"]))
    keep = [C.create_string_buffer(p, len(p)) for p in payloads]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, nh, nv, 1, 1, 0, 0, 0, 0, 1, 3)
    got, lock = {}, threading.Lock()

    def read_frame(frame, dst, nbytes):
        C.memmove(dst, keep[frame % len(keep)], nbytes)
        return 0

    def packet_done(frame, data, size):
        b = C.string_at(data, size)
        with lock:
            assert frame not in got
            got[frame] = b
        return 0

    st, _ = api.encode_sequence(cfg, n, read_frame, packet_done, batch=r.choice([0, 1, 3, 8, 16]), in_ring_frames=r.choice([0, 2, 3, 9]),
                                out_ring_bytes=r.choice([0, 1 << 16, 1 << 20]), readers=r.choice([1, 2, 5]), writers=r.choice([1, 3]),
                                lanes_per_device=r.choice([0, 1, 2]), copy_streams=copy_streams)
    assert st.frames == n and sorted(got) == list(range(n))
    p = ob.Params(w, h, pixfmt, nh, nv, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert got[i] == want[i % len(want)], f"seed {seed}: packet {i} differs from the oracle's"


@pytest.mark.parametrize("copy_streams,lanes", [(1, 1), (3, 1), (2, 2)])
def test_hd_sequence_with_real_transfers_equals_the_oracle(copy_streams, lanes, monkeypatch):
    """1920x1080 frames (12 MB payloads, ~11 MB packets): uploads, kernels and downloads really overlap here; a download ring of two
    chunks and eight upload slots keep every recycling path busy.  Packets are the oracle's, in any stream / lane arrangement."""
    w, h, pixfmt, n, distinct = 1920, 1080, synth.PIX_RGB16_BE, 60, 6
    payloads, line_bytes = _sequence(w, h, pixfmt, distinct)
    keep = [C.create_string_buffer(p, len(p)) for p in payloads]
    cfg = api.Ffv1Config(w, h, pixfmt, line_bytes, 4, 4, 1, 1, 0, 0, 0, 0, 1, 3)
    got, lock = {}, threading.Lock()

    def read_frame(frame, dst, nbytes):
        C.memmove(dst, keep[frame % distinct], nbytes)
        return 0

    def packet_done(frame, data, size):
        b = C.string_at(data, size)
        with lock:
            got[frame] = b
        return 0

    st, _ = api.encode_sequence(cfg, n, read_frame, packet_done, batch=7, in_ring_frames=8, out_ring_bytes=1 << 20, readers=4, writers=3, lanes_per_device=lanes, copy_streams=copy_streams)
    assert st.frames == n and sorted(got) == list(range(n))
    p = ob.Params(w, h, pixfmt, 4, 4, 1, 1)
    want = [ob.encode_payload(p, pl, line_bytes) for pl in payloads]
    for i in range(n):
        assert got[i] == want[i % distinct], f"packet {i} differs from the oracle's"


def test_bench_gpus_2_on_one_gpu_prints_both_multi_gpu_records(built, tmp_path):
    """`bench.py --gpus 2` on a box with one GPU (what the driver's N > 1 command turns into here): the headline on the device, then the two multi-GPU
    records over ALIASES of it -- `jobs_side_by_side` (N rcgpu-ffmpeg processes, N Matroska files: how a node pays off under the single-file
    ceiling) and `single_process_sharding` (one sequence, a lane per device, one placer) -- every packet equal to the N = 1 run's."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "512", "--height", "270", "--slices", "4", "--batch", "32",
                        "--host-frames", "128", "--e2e-frames", "128", "--legs", "host,e2e", "--no-verify", "--detail-out", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    # the line the driver parses is compact (one object under 4 KB: the contract's keys and the flat numbers); the records are in the detail file
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) < 4096 and line["roofline"]["kernel"] and line["detail"] == "detail.json" and len(line.get("step_ms", [])) <= 1
    d = json.load(open(tmp_path / "detail.json"))
    assert line["config"]["jobs_side_by_side_fps"] == d["jobs_side_by_side"]["value"] and line["config"]["single_process_sharding_fps"] == d["single_process_sharding"]["value"]
    j = d["jobs_side_by_side"]
    assert j["jobs"] == 2 and len(j["per_job"]) == 2 and all(x["all_blocks_identical_to_the_n1_runs_packets"] and x["frames"] == j["frames_per_job"] for x in j["per_job"]), j
    s = d["single_process_sharding"]
    assert s["devices"] == 2 and s["packets_identical_to_device_resident_run"] and len(s["lanes"]) == 2 and s["pinned_rings_on_their_devices_nodes"], s
    assert d["config"]["jobs_side_by_side_fps"] == j["value"] and d["config"]["single_process_sharding_fps"] == s["value"]
