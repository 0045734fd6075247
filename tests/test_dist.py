"""The N>1 path of bench.py on CPU: two processes, gloo backend, --dry-run (no engine, no compute).
Checks the launcher contract (env rendezvous on 127.0.0.1), the barrier / max-over-ranks reduction and
that exactly one JSON line comes from rank 0 with the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "5", "--warmup", "1", "--backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak" and out["unit"] == "pairs/s"
    # rank 1 sleeps 2 ms per step: the max over ranks (>= 10 ms for 5 steps) must be what is reported
    assert out["ms_per_step"] >= 2.0
    assert abs(out["value"] - 2 * 5 / (out["ms_per_step"] * 5 / 1e3)) < 1e-6 * out["value"]
    # the PCIe-inclusive frame-sequence leg runs on EVERY rank and is reduced like `value`: frames of all ranks over the
    # max elapsed (rank 1 "takes" 2 ms per frame in the dry run)
    hs = out["host_sequence_all_ranks"]
    assert hs["ranks"] == 2 and hs["frames_per_rank"] == 64 and hs["ms_per_frame_per_rank"] >= 2.0
    assert abs(hs["value"] - 2 * 64 / (hs["ms_per_frame_per_rank"] * 64 / 1e3)) < 1e-6 * hs["value"]
    # an N > 1 line says where the CPU leg is instead of leaving the key out
    assert out["cpu_baseline"] is None and "N = 1" in out["cpu_baseline_reason"]


def test_failing_tiled_leg_does_not_cost_the_line():
    """A multi-rank default run measures configs[3] in child processes started before the ranks touch the GPU
    (bench.tiled_children).  Here there is no GPU, so the children fail: the run must still exit 0 with ONE line
    that carries the headline fields and an error entry for the leg."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1", "--backend", "gloo", "--dry-run", "--rehearse-tiled-leg", "--tiled-timeout", "120"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "error" in out["tiled_4096x2160"], out["tiled_4096x2160"]


def test_shard_is_rank_local():
    sys.path.insert(0, ROOT)
    import bench
    # rank 3 of 8, one pair per step: its four resident pairs 12..15 in rotation, nobody else's
    assert bench.shard(3, 8, 6, 1) == [[12], [13], [14], [15], [12], [13]]
    # a batch repeats the rank's own four pairs (round 6: the batch legs match the headline's pairs, not 4 x nb others)
    assert bench.shard(3, 8, 2, 6) == [[12, 13, 14, 15, 12, 13], [13, 14, 15, 12, 13, 14]]
    mine = {i for step in bench.shard(3, 8, 8, 2) for i in step}
    other = {i for r in (2, 4) for step in bench.shard(r, 8, 8, 2) for i in step}
    assert mine == set(range(12, 16)) and not (mine & other)


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (what the driver's N=1..8 sweep issues when it does not wrap the
    command in torch.distributed.run): the script starts its two ranks itself, rank 0's line says n_gpus 2 and the
    process group the timed barrier saw had 2 ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend",
           "gloo", "--dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 4
    assert out["ms_per_step"] >= 2.0  # rank 1's 2 ms per step: the max over ranks
    # --gpus 1 stays a single process without a process group
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0",
                         "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    out1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert out1["n_gpus"] == 1 and out1["rccl_ranks"] == 1


def test_bench_gpus_flag_reports_a_failed_rank(tmp_path):
    """A rank that dies must end the run with a non-zero exit code instead of leaving the others at the barrier."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PM_BENCH_FAIL_RANK"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--backend",
           "gloo", "--dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank(s) failed" in r.stderr


@pytest.mark.gpu
def test_bench_with_a_one_rank_rccl_group_on_the_gpu():
    """The nccl (= RCCL) side of bench.py on hardware, as far as a one-GPU box allows: a process group of ONE rank
    (PM_BENCH_FORCE_DIST=1), so the barriers around the timed region and the max-over-ranks all-reduce run real RCCL
    kernels on RCCL's own stream beside the engine's streams.  The line must carry the headline fields, the world size
    the barrier saw, and a rate that shows the two views did not end up on one hardware queue."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--host-pairs", "16", "--no-side-legs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["unit"] == "pairs/s"
    assert out["check"]["deterministic_across_steps"] is True
    assert out["value"] > 150.0, out["value"]  # (one queue for both views reads ~275 on an MI355X; far lower = broken)
    # the PCIe-inclusive frame sequence of configs[2] ran on the rank, between RCCL barriers, reduced over the group
    hs = out["host_sequence_all_ranks"]
    assert hs["ranks"] == 1 and hs["frames_per_rank"] == 16 and hs["value"] > 150.0, hs
