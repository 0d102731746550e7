"""N > 1 path on CPU: world-size-2 gloo processes exercise the flat-gradient all-reduce,
the parameter broadcast and the env:// rendezvous that bench.py uses under
torch.distributed.run (the loss kernels themselves need a GPU and are shard-local)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from taiyaki_amd import models, parallel

    rank, local, world = parallel.init_from_env(backend="gloo")
    assert world == 2 and dist.get_world_size() == 2
    torch.manual_seed(100 + rank)                 # different init per rank on purpose
    net = models.mLstm_flipflop(size=16, stride=5)
    parallel.broadcast_parameters(net)            # now identical to rank 0
    flat0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref = flat0.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat0, ref), "parameter broadcast failed"

    arena = parallel.FlatGradArena(net)
    assert arena.flat.numel() == sum(p.numel() for p in net.parameters() if p.requires_grad)
    # each rank its own shard of reads; plain torch loss (kernels are GPU-only)
    torch.manual_seed(7 + rank)
    x = torch.randn(200, 3, 1)
    arena.zero()
    net(x).square().mean().backward()
    local_grad = arena.flat.clone()
    assert all(p.grad.data_ptr() >= arena.flat.data_ptr() for p in arena.params), "grads left the arena"
    arena.allreduce_async()
    arena.finish()
    both = [torch.zeros_like(local_grad) for _ in range(2)]
    dist.all_gather(both, local_grad)
    expect = (both[0] + both[1]) / 2
    assert torch.allclose(arena.flat, expect, rtol=1e-6, atol=1e-8), "all-reduce mean mismatch"
    # one optimiser step keeps the replicas identical
    opt = torch.optim.AdamW(arena.params, lr=1e-3)
    opt.step()
    flat1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref = flat1.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat1, ref), "replicas diverged after the step"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
''') % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_flat_allreduce_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-3000:])
        assert "rank %d ok" % rank in out


def test_torchrun_env_rendezvous_single_process():
    """bench.py's rendezvous helper is a no-op at world size 1 (the default `python bench.py`)."""
    from taiyaki_amd import parallel
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert parallel.init_from_env() == (0, 0, 1)
