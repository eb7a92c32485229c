"""Multi-process (world_size 2, gloo, CPU) test of the bucketed gradient reducer that the
N-GPU benchmark uses over RCCL: per-flow flat buckets whose slices are the parameters' .grad,
async all-reduce launched from grad hooks, mean over ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Flow(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(c, c) * 0.3)
        self.b = torch.nn.Parameter(torch.zeros(c))
        self.frozen = torch.nn.Parameter(torch.ones(c), requires_grad=False)

    def forward(self, z):
        return torch.tanh(z @ self.w + self.b) * self.frozen


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.flows = torch.nn.ModuleList([_Flow(6) for _ in range(3)])
        self.context_lstm = torch.nn.LSTM(6, 3, batch_first=True, bidirectional=True)

    def forward(self, x):
        y, _ = self.context_lstm(x)
        for f in self.flows:
            y = f(y)
        return (y ** 2).mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rad_mmm_amd.ddp import BucketedGradReducer, broadcast_module_state, default_bucket_key
    torch.manual_seed(100 + rank)           # different init per rank on purpose
    model = _Toy()
    broadcast_module_state(model, 0)
    red = BucketedGradReducer(model)
    keys = [b["key"] for b in red.buckets]
    assert keys == ["flows.0.lo", "flows.1.lo", "flows.2.lo", "misc"], keys
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 4, 5, 6, generator=g)
    ok = True
    for it in range(2):                      # twice: prepare() must re-arm the hooks
        red.prepare()
        loss = model(data[rank])
        loss.backward()
        red.finish()
        # reference: mean over ranks of the local gradients, computed serially on every rank
        ref = [torch.zeros_like(p) for p in model.parameters() if p.requires_grad]
        for r in range(world):
            m2 = _Toy()
            m2.load_state_dict(model.state_dict())
            m2(data[r]).backward()
            for a, p in zip(ref, [p for p in m2.parameters() if p.requires_grad]):
                a += p.grad / world
        for a, p in zip(ref, [p for p in model.parameters() if p.requires_grad]):
            ok = ok and torch.allclose(p.grad, a, rtol=1e-5, atol=1e-7)
        # .grad tensors are views into the flat buckets
        for b in red.buckets:
            off = 0
            for p in b["params"]:
                ok = ok and p.grad.data_ptr() == b["flat"].data_ptr() + 4 * off and off % 4 == 0
                off += (p.numel() + 3) & ~3                      # 16-byte aligned slots (ddp.slot_numel)
    # identical parameters on every rank after broadcast
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    # the saturation flag words travel as a bitwise OR (GradScale, strict + distributed) -- through MAX over 0/1 words,
    # because RCCL has no BOR reduction
    from rad_mmm_amd.ops import _all_reduce_or
    fl = torch.tensor([[1, 4], [2, 4 | 128]][rank], dtype=torch.int32)
    ok = ok and _all_reduce_or(fl).tolist() == [3, 132] and fl.tolist() == [[1, 4], [2, 132]][rank]
    # the step's loss terms mean-reduced across ranks in ONE collective (the reference: one per term,
    # tts_lightning_modules.py:746-749); python numbers among the values, names packed in sorted order on every rank
    from rad_mmm_amd.ddp import reduce_loss_dict
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        ld = ({"loss_mel": (torch.tensor(1.0 + rank), 1.0), "loss_ctc": (torch.tensor(0.5 * (rank + 1)), 0.1),
               "binarization_loss": (0.0, 1.0)} if rank == 0 else
              {"binarization_loss": (0.0, 1.0), "loss_ctc": (torch.tensor(0.5 * (rank + 1)), 0.1),
               "loss_mel": (torch.tensor(1.0 + rank), 1.0)})                       # another insertion order on rank 1
        h = reduce_loss_dict(ld)
        got = {k: float(v) for k, v in h.wait().items()}
        got2 = {k: float(v) for k, v in reduce_loss_dict(ld, async_op=False).wait().items()}
    finally:
        dist.all_reduce = real
    want = {"loss_mel": 1.5, "loss_ctc": 0.75, "binarization_loss": 0.0}
    ok = ok and len(calls) == 2 and got == want and got2 == want and float(ld["loss_mel"][0]) == 1.0 + rank
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_reducer_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)], res


def test_loss_dict_reduce_without_a_process_group_is_local():
    from rad_mmm_amd.ddp import reduce_loss_dict
    h = reduce_loss_dict({"b": (torch.tensor(2.0), 1.0), "a": 3, "c": torch.tensor([4.0])})
    assert {k: float(v) for k, v in h.wait().items()} == {"a": 3.0, "b": 2.0, "c": 4.0}
    assert reduce_loss_dict({}).wait() == {}


def test_bucket_key():
    from rad_mmm_amd.ddp import default_bucket_key
    assert default_bucket_key("flows.3.coupling_tfn.affine_param_predictor.start.weight_v") == "flows.3.lo"
    assert default_bucket_key("flows.3.coupling_tfn.affine_param_predictor.in_layers.1.conv.weight_g") == "flows.3.lo"
    assert default_bucket_key("flows.3.coupling_tfn.affine_param_predictor.res_skip_layers.2.bias") == "flows.3.hi"
    assert default_bucket_key("flows.3.coupling_tfn.affine_param_predictor.end.weight") == "flows.3.hi"
    assert default_bucket_key("decoder.flows.11.invtbl_conv.lower") == "decoder.flows.11.lo"   # reducer around the whole step
    assert default_bucket_key("f0_predictor.feat_pred.lstm.weight_hh_l0") == "misc"
    assert default_bucket_key("context_lstm.weight_ih_l0") == "misc"


def test_collective_cu_reservation_reaches_the_library():
    """ddp.reserve_collective_cus() (what bench.py calls for N > 1 before RCCL starts) pins RCCL's channel count and makes
    libradmmm_hip.so size its GEMM grids for the remaining CUs (radmmm_gemm_cu_slots reads RADMMM_GEMM_CUS once per
    process, hence the child process)."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "for k in ('NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'RADMMM_GEMM_CUS'): os.environ.pop(k, None)\n"
            "from rad_mmm_amd.ddp import reserve_collective_cus, RCCL_CUS\n"
            "reserve_collective_cus()\n"
            "from rad_mmm_amd._lib import lib\n"
            "print(os.environ['NCCL_MAX_NCHANNELS'], os.environ['NCCL_MIN_NCHANNELS'], os.environ['RADMMM_GEMM_CUS'], "
            "lib.radmmm_gemm_cu_slots(), RCCL_CUS)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    mx, mn, cus, slots, n = out.stdout.split()[-5:]
    assert mx == mn == n == "16" and cus == "240" and slots == "240"
