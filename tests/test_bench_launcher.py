"""CPU: bench.py starts its own ranks when `--gpus N` is invoked plainly (no torch.distributed.run around it), relays
rank 0's one JSON line, and refuses within seconds when the host has fewer GPUs.  The rehearsal (`--dry-run`) runs the
same launcher and the same pack -> all_gather -> count plumbing over gloo, without touching a GPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_plain_invocation_with_two_gpus_launches_two_ranks():
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=_env(FFS_BENCH_ASSUME_GPUS="2"), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["dry_run"] is True and d["steps"] == 3
    assert time.time() - t0 < 200


def test_fewer_gpus_than_asked_for_fails_fast_and_cleanly():
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(FFS_BENCH_ASSUME_GPUS="1"),
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    assert "--gpus 8" in p.stderr and "1 GPU(s) visible" in p.stderr
    assert p.stdout.strip() == ""
    assert time.time() - t0 < 30


def test_under_torchrun_style_environment_it_is_a_rank_not_a_launcher():
    # WORLD_SIZE set => no children are started: a one-rank dry run prints its line directly
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"],
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1


def test_a_failing_rank_takes_the_run_down_with_its_code():
    # rank 1 dies before the rendezvous; rank 0 would wait for it for ever: the launcher stops it and reports rank 1's code
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"],
                       env=_env(FFS_BENCH_ASSUME_GPUS="2", FFS_BENCH_DRYRUN_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 7 and p.stdout.strip() == ""
    assert "failed" in p.stderr
    assert time.time() - t0 < 60


def test_three_ranks_gather_rows_with_an_empty_rank_and_the_padded_partner():
    # --gather rows (default): counts + send/recv to rank 0, rank 1 has no rows; --gather padded: the all_gather of blocks
    for extra in ([], ["--gather", "padded"]):
        p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-run", *extra], env=_env(FFS_BENCH_ASSUME_GPUS="3"),
                           capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == 3 and d["n_ranks_seen"] == 3, d     # (the empty rank is seen too: its tag travels with its count)
