import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the tests force kernel choices (tile sizes, product schemes on tiny batches, the persistent LSTM): those switches are
    # honoured only under RADMMM_DEBUG=1 (rad_mmm_amd/_lib.py debug_env, csrc/error.cpp), set before the library loads
    os.environ.setdefault("RADMMM_DEBUG", "1")
    # the HIP library is a build artefact (git-ignored): cross-compile it if this is a fresh tree
    if not os.path.exists(os.path.join(ROOT, "rad_mmm_amd", "libradmmm_hip.so")):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "rad_mmm_amd", "csrc", "build.sh")])


def host_threads() -> int:
    """CPU threads this process may really use (affinity mask capped by the cgroup CPU quota).  torch's default is the
    host's core count: on a box whose container is limited to a few cores the CPU oracle then runs oversubscribed and
    5-10x slower -- round 3's GPU suite spent 600 of its 1170 s there."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_sessionstart(session):
    import torch
    torch.set_num_threads(host_threads())


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def sub(d, prefix):
    """{k[len(prefix):]: v} for keys starting with prefix."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
