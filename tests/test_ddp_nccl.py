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


def test_bench_control_flow_at_world_size_two_on_one_gpu():
    """The driver's N > 1 runs are the first time two ranks meet, and a collective that only rank 0 issues (an extra
    measured step, a flag all-reduce inside a rank-0 block) hangs them: round 4 found two by reading.  This runs the
    benchmark's whole N = 2 control flow on ONE GPU -- both ranks on device 0, gradients reduced by gloo through the host
    (RCCL refuses two ranks on a device) -- under a timeout: timed loop, exposed-communication report, per-bucket all-reduce
    timing, in-step launch timing, saturation report, JSON line from rank 0."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "8", "--frames", "400"], capture_output=True, text=True, timeout=600,
                       env=_env(RADMMM_BENCH_SHARE_GPU="1", RADMMM_BENCH_BACKEND="gloo", RADMMM_CHECK_SATURATION="1"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 16
    d = res["distributed"]
    assert d["backend"] == "gloo" and d["rccl_world_size"] == 2 and d["rccl_allreduce_of_ones"] == 2.0
    assert len(d["allreduce_alone_per_bucket"]) == d["gradient_buckets"] and res["saturation"]["nonfinite_passes"] == 0
    assert res["roofline"]["avg_launch_ms"] > 0
    # the loss terms of the last timed step, mean-reduced over the ranks in one collective per step -- rank 0's own
    # NLL (its own utterances) differs from the mean over both ranks
    assert res["loss_mel_global"] is not None and set(res["loss_terms_global"]) >= {"loss_mel", "loss_prior_mel"}
    assert abs(res["loss_mel_global"] - res["loss_mel"]) > 1e-6 * abs(res["loss_mel"])


def _world2(*extra, port):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_ddp_world2.py"), *extra],
                          capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)


def test_gradient_exchange_values_at_world_size_two():
    """VALUES of the gradient exchange with two ranks on the real decoder (tests/_ddp_world2.py): every .grad after
    finish() equals the mean of the two ranks' single-rank gradients to <= 1e-6 (measured: bit-equal), two steps in a row,
    with direct sinks and the early bucket start taken; then one spline flow with synchronised masked batch-norm against
    the single-process run on the concatenated batch.  Reference: configs/RADMMM_train_config.yaml:28 (`strategy: ddp`),
    maskedbatchnorm1d.py:88-95."""
    r = _world2(port=29631)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    assert r.stdout.count("DDP_WORLD2_AFFINE_OK") == 2 and r.stdout.count("DDP_WORLD2_SPLINE_SYNCBN_OK") == 2, r.stdout[-3000:]


def test_gradient_exchange_check_detects_a_premature_all_reduce():
    """Control for the test above: a bucket announced final before its gradients are written (what a wrong
    notify_grads_final would do, invisible at world size 1) makes the same comparison fail."""
    r = _world2("--negative", port=29633)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    assert r.stdout.count("DDP_WORLD2_NEGATIVE") == 2, r.stdout[-3000:]


def test_torch_ddp_wrapper_gives_the_reducers_gradients():
    """The path Lightning's `strategy: ddp` takes (configs/RADMMM_train_config.yaml:28): the decoder wrapped in stock
    torch.nn.parallel.DistributedDataParallel, two ranks on their own utterances, against a twin through
    BucketedGradReducer -- every parameter's gradient equal to <= 1e-6 (tests/_ddp_torch_world2.py).  Backs the sentence
    "plain torch DDP also works" in INTEGRATION.md."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "tests", "_ddp_torch_world2.py")],
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    assert r.stdout.count("DDP_TORCH_WRAPPER_OK") == 2, r.stdout[-3000:]
