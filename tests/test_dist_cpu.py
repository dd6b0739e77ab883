"""CPU: the sharding of the product and the plumbing of the N>1 bench.  Frames shard with no data-path collective: rcgpu_sequence_plan is the
plan rcgpu_ffv1_encode_sequence follows inside one process (a lane per device, rawcooked_amd/csrc/pipeline.hip) and needs no device; the
bench's ranks (one per GPU, as the driver launches them) use the process group for the barrier and the max-over-ranks timing only.  World
size 2 over gloo exercises exactly those calls, with every rank taking the frames the plan gives "its" lane."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_products_shard_plan_covers_every_frame_once_in_order():
    sys.path.insert(0, ROOT)
    from rawcooked_amd import api
    for n in (0, 1, 7, 64, 1000, 2000):
        for lanes in (1, 2, 3, 8):
            for batch in (1, 5, 336):
                lane, bat = api.sequence_plan(n, batch, lanes)
                # the lanes the job really gets: a device per 8 frames at most (pipeline::prepare's own rule -- 20 frames on 8 devices run on 3)
                L = api.sequence_plan_lanes(n, lanes)
                assert L == max(1, min(lanes, (n + 7) // 8))
                assert len(lane) == n and all(0 <= x < L for x in lane)
                assert bat == [i // batch for i in range(n)]                     # batches are runs of consecutive frames ...
                assert lane == [(i // batch) % L for i in range(n)]             # ... dealt to the lanes in turn (batch b -> lane b mod L)
    assert api.sequence_plan_lanes(20, 8) == 3 and api.sequence_plan_lanes(20, 8, 2) == 3 and api.sequence_plan_lanes(2000, 8, 2) == 16
    assert set(api.sequence_plan(20, 4, 8)[0]) == {0, 1, 2}
    lane, _ = api.sequence_plan(10, 2, 2)
    assert [i for i in range(10) if lane[i] == 0] == [0, 1, 4, 5, 8, 9] and [i for i in range(10) if lane[i] == 1] == [2, 3, 6, 7]
    # BASELINE config 4: 2000 frames over 8 GPUs in batches of 80 -> 25 batches, lanes 0..7 then 0 again; no lane idles while another has two more
    lane, bat = api.sequence_plan(2000, 80, 8)
    per_lane = [sum(1 for x in lane if x == k) for k in range(8)]
    assert max(per_lane) - min(per_lane) <= 80 and sum(per_lane) == 2000
    with __import__("pytest").raises(api.RcgpuError):
        api.sequence_plan(10, 0, 2)


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        import bench_dist as rdist
        from bench_dist import max_over_ranks
        from rawcooked_amd import api
        # the same calls bench.py makes at N > 1, with gloo in place of RCCL
        d2 = rdist.init(int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), 0, "gloo", torch.device("cpu"))
        assert d2 is dist
        r = dist.get_rank()
        lane_of, _ = api.sequence_plan(10, 2, 2)          # the product's plan: this rank plays lane r
        mine = [i for i in range(10) if lane_of[i] == r]
        rdist.barrier(dist)
        t = max_over_ranks(dist, 1.0 + r, torch.device("cpu"))
        calls = []
        dt = rdist.timed_steps(dist, torch.device("cpu"), lambda: (calls.append(1), time.sleep(0.05 * (r + 1))), 3, 2, lambda: None)
        assert len(calls) == 5 and dt >= 0.29, (calls, dt)          # 2 warm-up + 3 timed steps; the slower rank's 3 x 0.1 s
        n = torch.tensor([len(mine)]); dist.all_reduce(n)          # test-side bookkeeping only: the encode path has no collective
        assert t == 2.0 and int(n) == 10, (t, int(n))
        rows = rdist.gather_floats(dist, [r, 2.5 * r], torch.device("cpu"))      # bench.py's per-rank host_pipeline rows
        assert rows == [[0.0, 0.0], [1.0, 2.5]], rows
        assert rdist.gather_floats(None, [3, 4], torch.device("cpu")) == [[3.0, 4.0]]
        rdist.finish(dist)
        print("rank", r, "ok", mine)
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.count("ok") == 2, r.stdout + r.stderr


def test_bench_command_line_is_the_contract():
    """`python bench.py --gpus N --steps K --warmup W` is what the driver runs: the options exist, and the file at least parses here."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for opt in ("--gpus", "--steps", "--warmup", "--legs", "--mode"):
        assert opt in r.stdout, opt


def _canned_result(n_gpus):
    """What bench.py's main() holds at the end of a run, with the prose and the per-leg records a real run carries (and more)."""
    prose = "a paragraph of explanation that belongs in bench_detail.json and not in the line the driver parses; " * 12
    r = {"metric": "4K-DCI 16-bit DPX->FFV1 frames/sec", "value": 701.5 * n_gpus, "unit": "frames/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 479.0,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic", "h2d_included": False,
         "config": {"workload": "BASELINE config 2: 4096x2160 RGB 16-bit BE DPX payload -> FFV1 v3 intra, slices=64 (8x8), coder=1 context=1 (ffmpeg level maps) slicecrc=1, content=film; " + prose,
                    "range_coder": "one lane per slice", "frames_per_step_per_gpu": 336, "parallelism": "frame-sharded x%d, no collective" % n_gpus, "steps_issued": prose,
                    "steps_issued_probe": {"run_on_ms_per_step": 502.3, "one_batch_at_a_time_ms_per_step": 512.3}, "packet_bytes_avg": 48847621, "compression_ratio": 0.9202,
                    "decisions_per_frame": 552785634, "verified_vs_oracle": True, "verified_by_reference": True, "device_error_flags": 0, "hbm_in_use_gb": 218.2, "packets_verified": prose,
                    **{"some_leg_%d_fps" % i: 100.0 + i for i in range(40)}},
         "roofline": {"bound": "hbm", "kernel": "k_resolve", "achieved": 73.992, "peak": 8000.0, "unit": "GB/s", "frac": 0.009249, "traffic": 31939375818, "traffic_stale": False,
                      "request_frac": 0.7201, "floor_ms": 8.916, "algorithmic_bytes_per_launch": 1070283707, "launch_ms": 14.465, "launches_per_step": 32,
                      "kernel_ms_per_step": {"k_model": 329.966, "k_resolve": 462.878, "k_rangecode": 437.875, "k_footer": 4.586, "k_scan": 0.007, "k_gather": 8.85},
                      "note": prose, "requests": {"ceiling_what": prose, "floor_what": prose}, "issue_frac": {"note": prose}},
         "cpu_baseline": {"value": 5.478, "unit": "frames/s", "cores": 16, "cores_busy": 16, "kind": "port", "sample": "16 processes x 2 frames of the same 4096x2160 RGB16 workload (32 distinct frames); " + prose},
         "kernel_metric": {"value": 610.0, "unit": "frames/s", "h2d_included": True, "frames": 1000, "what": prose},
         "host_pipeline": {"trace": prose, "per_rank": [{"rank": r_, "frames": 3840, "seconds": 6.5} for r_ in range(n_gpus)]},
         "e2e": {"timeline": [prose] * 40}, "check": {"roofline": {"note": prose}}, "configs": {"cfg4": {"what": prose}}, "long_sequence": {"timeline": [prose] * 60}}
    if n_gpus > 1:
        r["rccl"] = {"backend": "nccl", "world_size": n_gpus, "ranks_counted_by_all_reduce": n_gpus, "saw_every_rank": True, "used_for": prose}
        r["jobs_side_by_side"] = {"value": 1400.0, "jobs": n_gpus, "per_job": [{"job": j, "frames": 1000, "seconds": 5.0 + j} for j in range(n_gpus)], "what": prose}
        r["single_process_sharding"] = {"value": 260.0, "lanes": [{"device": j} for j in range(n_gpus)], "what": prose}
        r["config"].update(jobs_side_by_side_fps=1400.0, single_process_sharding_fps=260.0, rccl_ranks=n_gpus)
    return r


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_bench_line_is_compact_and_carries_the_contract(n_gpus):
    """The driver keeps the last 8 KB of stdout and parses the last line (round 5's 26 KB line left `parsed: null`): whatever a run
    measured, the line is one JSON object under 4 KB with the contract's keys, `roofline` and `cpu_baseline` as numbers and the
    per-step times; the prose and the per-leg records stay in the detail file."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    res = _canned_result(n_gpus)
    assert len(json.dumps(res)) > 30000                      # the record is far larger than the driver keeps
    steps = [479.0 + 0.37 * k for k in range(20)]
    line = bench.compact_line(res, steps)
    assert "\n" not in line and len(line.encode()) < 4096 == bench.LINE_LIMIT, len(line)
    d = json.loads(line)
    assert json.loads(json.dumps(d)) == d
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == res["value"] and d["n_gpus"] == n_gpus and d["ms_per_step"] == 479.0 and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("BASELINE config 2") and len(d["config"]["workload"]) <= 200 and "model" not in d["config"]
    assert all(not isinstance(v, (dict, list)) for v in d["config"].values())                    # flat: the driver keeps config values it can print
    assert d["config"]["frames_per_step_per_gpu"] == 336 and d["config"]["verified_vs_oracle"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "k_resolve" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and rf["traffic"] == 31939375818 and rf["launches_per_step"] == 32
    assert all(not isinstance(v, str) or k in ("bound", "kernel", "unit") for k, v in rf.items())   # numbers only
    cb = d["cpu_baseline"]
    assert cb["value"] == 5.478 and cb["cores"] == 16 and cb["kind"] == "port" and cb["unit"] == "frames/s" and 0 < len(cb["sample"]) <= 120
    assert d["step_ms"] == [round(x, 1) for x in steps] and d["detail"] == bench.DETAIL_FILE
    assert d["kernel_metric"] == {"value": 610.0, "unit": "frames/s", "h2d_included": True, "frames": 1000}
    if n_gpus > 1:
        assert d["config"]["rccl_ranks"] == n_gpus and d["config"]["jobs_side_by_side_fps"] == 1400.0
    # a real record of the last round (26.5 KB as printed then) comes out under the limit too
    real = os.path.join(ROOT, "profiles", "r05_bench_default.json")
    if os.path.exists(real):
        line = bench.compact_line(json.load(open(real)), steps)
        assert len(line) < 4096 and json.loads(line)["roofline"]["kernel"] == "k_resolve"
