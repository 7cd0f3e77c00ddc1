"""bench.py's own multi-rank launch path on CPU: `--gpus 2` with no WORLD_SIZE in the environment must start two ranks
itself (the parent never touches the GPU), rendezvous on 127.0.0.1, broadcast rank 0's prompt embeddings, time K steps
between barriers with the max over ranks, and relay rank 0's JSON line with n_gpus == 2 (round 1: `--gpus` was parsed
and ignored).  `--dry-run` replaces the GPU step by a sleep; gloo stands in for RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks():
    out = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"], {"VSD_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dry_run"] is True and out["steps"] == 5
    cs = out["prompt_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1] and cs[0] != 0.0  # rank 1 received rank 0's embeddings
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert out["ms_per_step"] >= 2.0


def test_single_rank_default_and_torchrun_environment():
    out = _run(["--steps", "3", "--dry-run"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1
    # the driver's form: the environment of torch.distributed.run is already there -> this process is a rank, not a launcher
    out = _run(["--gpus", "1", "--steps", "3", "--dry-run"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29731"})
    assert out["n_gpus"] == 1


def test_a_rank_that_dies_ends_the_run_at_once_and_leaves_no_process_behind():
    """a rank other than 0 that dies at start-up used to leave rank 0 in the rendezvous until the process-group
    timeout (minutes) before the launcher said anything.  The launcher polls every child: it exits non-zero within seconds
    and stops the survivors."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"VSD_DIST_BACKEND": "gloo", "VSD_DRYRUN_FAIL_RANK": "1"})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "rank 1 exited with 7" in (p.stderr + p.stdout), (p.stdout, p.stderr[-500:])
    assert time.time() - t0 < 60


def test_bench_refuses_to_run_the_product_path_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)


import pytest


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_box_end_to_end():
    """The multi-rank GPU path for real (launcher -> 2 ranks -> prompt broadcast -> engines -> barrier / max-over-ranks ->
    rank 0's line).  The test box has ONE GPU, so both ranks share cuda:0 and gloo carries the collectives
    (VSD_SHARE_GPU / VSD_DIST_BACKEND exist for exactly this; the driver's 8-GPU run uses RCCL and one GPU per rank)."""
    out = _run(["--gpus", "2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-extras"],
               {"VSD_DIST_BACKEND": "gloo", "VSD_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["value"] > 10.0 and out["scaling"] == "weak"
    assert out["config"]["sharding"].startswith("round-robin frames over 2")
