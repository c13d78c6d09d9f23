"""`python bench.py --gpus N` must start N ranks itself: run plainly - the way the driver runs N = 1 - it
used to read WORLD_SIZE only and measured ONE GPU whatever --gpus said.  These tests run the real bench.py as a subprocess with
its `--plumbing-only` stand-in model (CPU ranks, gloo, no measurement): the self-launch through torch.distributed.run, the
process group, the scatter / gather of parler_tts_amd.distributed and rank 0's JSON line are the product code paths."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PTTS_SHARE_GPU", "PTTS_DIST_BACKEND")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)


def test_gpus_2_starts_two_ranks_and_ends_with_rank0_json_line():
    r = _run(["--gpus", "2", "--plumbing-only", "--batch", "3", "--max-length", "12", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    line = json.loads(last)                                   # the JSON line is the LAST line of stdout
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 6
    dp = line["data_parallel"]
    assert dp["ranks"] == 2 and dp["backend"] == "gloo"
    assert set(dp["per_rank_min_max"]) >= {"scatter_ms_last_pass", "generate_ms_last_pass", "gather_ms_last_pass"}
    assert dp["equals_single_gpu"] is True                    # rank 1's slice re-generated on rank 0 gives the gathered ids
    assert line["steps"] == 2 and line["warmup"] == 1 and line["passes_bit_identical_rank0"] is True
    # a rehearsal can not be mistaken for a measurement
    assert line["value"] == 0.0 and "invalid" in line and "roofline" not in line and line["data"].startswith("none")
    assert sum('"metric"' in ln for ln in r.stdout.splitlines()) == 1       # one line, from rank 0


def test_more_ranks_than_devices_is_an_error_not_a_one_gpu_measurement():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("needs a host with fewer than 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert '"metric"' not in r.stdout and "refusing to measure fewer ranks" in r.stderr


def test_launcher_rank_count_must_agree_with_the_flag():
    r = _run(["--gpus", "2", "--plumbing-only"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "must agree" in r.stderr and '"metric"' not in r.stdout
    # --gpus 1 inside a 2-rank launch is refused as well (it used to be accepted silently when WORLD_SIZE was 1)
    r = _run(["--gpus", "1", "--plumbing-only"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "must agree" in r.stderr


def test_eight_rank_rehearsal_of_the_scale_run():
    """The driver's SCALE run is `--gpus 8` on a node the builder never sees: the same launcher, process group, scatter of the id
    slices and padded gather with 8 CPU ranks (only 2 ranks were rehearsed before)."""
    r = _run(["--gpus", "8", "--plumbing-only", "--batch", "2", "--max-length", "12", "--steps", "1", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["config"]["global_batch"] == 16
    dp = line["data_parallel"]
    assert dp["ranks"] == 8 and dp["equals_single_gpu"] is True
    assert sum('"metric"' in ln for ln in r.stdout.splitlines()) == 1


def test_a_rank_that_computes_something_else_invalidates_the_line():
    """`data_parallel.equals_single_gpu` is a check, not a constant: a rank 1 whose codec ids differ from what rank 0 computes for
    the same slice must turn the line invalid (SURVEY section 4 test (5), inside the N > 1 line)."""
    r = _run(["--gpus", "2", "--plumbing-only", "--batch", "3", "--max-length", "12", "--steps", "1", "--warmup", "0"],
             env_extra={"PTTS_PLUMBING_CORRUPT_RANK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert line["data_parallel"]["equals_single_gpu"] is False and "differ from a single-GPU generate()" in line["invalid"]
