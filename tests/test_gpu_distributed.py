"""The data-parallel plumbing on the REAL backend: a world-size-1 RCCL group on cuda:0 in a child process (one GPU per box here;
RCCL refuses two ranks on one device, so rank counts > 1 are covered by the gloo tests and by the driver's SCALE run)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_collective_of_the_dp_path_runs_on_rccl_world_size_1():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on an MI355X box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_rccl_child.py")], capture_output=True, text=True, env=env,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["12", "7"], ["8", "7", "generate"], ["10", "7", "bf16"], ["8", "7", "fused"]],
                         ids=["teacher_forced", "generate_api", "bf16", "fused_launch"])
def test_randomised_parity_sweep(args):
    """tools/fuzz_parity.py in a child process: random tiny architectures and call shapes the fixtures do not cover (codebooks 1 .. 32,
    batch 1 .. 40, prompts across the 64-token KV page, ragged masks; with `generate`: the public call, free-running, live EOS heads,
    guidance) - fp32 engine vs the oracle; `fused` (round 5): decoders wide enough for the fused QKV + self-attention launch, batch up
    to three m-tiles, against the two launches of rounds 1-4, bit for bit.  A fixed seed: the same cases on every run."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on an MI355X box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py")] + args, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
