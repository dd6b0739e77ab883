"""CPU: the sharding of the product and the plumbing of the N>1 bench.  Frames shard with no data-path collective: rcgpu_sequence_plan is the
plan rcgpu_ffv1_encode_sequence follows inside one process (a lane per device, rawcooked_amd/csrc/pipeline.hip) and needs no device; the
bench's ranks (one per GPU, as the driver launches them) use the process group for the barrier and the max-over-ranks timing only.  World
size 2 over gloo exercises exactly those calls, with every rank taking the frames the plan gives "its" lane."""
import os
import subprocess
import sys
import textwrap

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
