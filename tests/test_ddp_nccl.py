"""RCCL on the GPU box (one GPU, so world size 1): the gradient reducer with direct sinks under an active "nccl"
process group, and bench.py starting its own ranks.  Both run in child processes (process-group state and a possible
RCCL hang stay out of the pytest process; each child has its own timeout)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.update(extra)
    return env


def test_decoder_reducer_with_sinks_under_nccl_world1_is_bit_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_world1.py")], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the driver's scaling run calls it that
    way); RADMMM_BENCH_SPAWN=1 takes the same path at N = 1: torch.distributed.run -> one rank -> nccl process group ->
    bucketed all-reduce inside the timed step -> rank 0 prints the JSON line last."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-throughput-mode"], capture_output=True, text=True, timeout=900,
                       env=_env(RADMMM_BENCH_SPAWN="1"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["value"] > 0
    assert res["config"]["parallelism"] == "dp1" and res["distributed"]["backend"] == "nccl"
