"""RCCL on the one-GPU box: the backend the N > 1 path uses (torch.distributed "nccl" = RCCL on ROCm) initialised with a
single rank, its collectives and the grouped send / receive that `sharding.gather_features` is built from, executed on
the card.  The multi-rank data path itself is covered by the gloo tests (tests/test_sharding_gloo.py, test_sweep.py);
what this adds is that the RCCL library loads, builds a communicator and runs kernels in this image."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    y = x.clone()
    dist.all_reduce(y)
    assert torch.equal(y, x)
    g = torch.empty_like(x)
    dist.all_gather_into_tensor(g, x)
    assert torch.equal(g, x)
    # the gather-v's primitive: grouped isend / irecv (ncclSend / ncclRecv inside one group), here rank 0 to itself
    out = torch.empty_like(x)
    for r in dist.batch_isend_irecv([dist.P2POp(dist.irecv, out, 0), dist.P2POp(dist.isend, x, 0)]):
        r.wait()
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    import importlib
    sh = importlib.import_module("hts-train-world_amd.sharding")
    outs, counts = sh.gather_features([x[:1000], x[:2000].reshape(1000, 2)], [400, 600])
    assert counts == [[400, 600]] and torch.equal(outs[1], x[:2000].reshape(1000, 2))
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_OK", torch.cuda.nccl.version())
""") % ROOT


@pytest.mark.gpu
def test_rccl_initialises_and_runs_on_one_rank():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
