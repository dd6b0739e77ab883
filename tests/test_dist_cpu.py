"""CPU, world_size 2 over gloo: the N>1 bench path shards frames with no data-path collective; only the barrier and the
max-over-ranks timing use the process group (bench.py).  This exercises exactly those two calls plus the shard plan."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_covers_every_frame_once():
    sys.path.insert(0, ROOT)
    from rawcooked_amd.dist import shard_frames
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += shard_frames(n, r, world, batch=5)
            assert sorted(seen) == list(range(n))
    assert shard_frames(10, 0, 2, batch=2) == [0, 1, 4, 5, 8, 9] and shard_frames(10, 1, 2, batch=2) == [2, 3, 6, 7]


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from rawcooked_amd import dist as rdist
        from rawcooked_amd.dist import shard_frames, max_over_ranks
        # the same calls bench.py makes at N > 1, with gloo in place of RCCL
        d2 = rdist.init(int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), 0, "gloo", torch.device("cpu"))
        assert d2 is dist
        r = dist.get_rank()
        mine = shard_frames(10, r, 2, batch=2)
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
