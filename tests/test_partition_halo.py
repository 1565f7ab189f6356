"""Multi-GPU path on CPU: Cartesian partition geometry, and a world_size-2 gloo rehearsal of the
face-trace halo exchange (the same HaloExchange code runs over RCCL on the GPUs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_grids_and_neighbours():
    from exahype_amd.solvers import CartesianPartition
    assert CartesianPartition(8, 0, 3).pdims == [2, 2, 2]
    assert CartesianPartition(4, 0, 3).pdims == [2, 2, 1]
    assert CartesianPartition(2, 0, 3).pdims == [2, 1, 1]
    assert CartesianPartition(1, 0, 3).pdims == [1, 1, 1]
    p = CartesianPartition(8, 5, 3)
    assert p.coords == [1, 0, 1] and p.neighbour(0, +1) == 1 and p.neighbour(2, -1) == 4 and p.neighbour(1, +1) == 7
    seen = {CartesianPartition(8, r, 3).rank_of(CartesianPartition(8, r, 3).coords) for r in range(8)}
    assert seen == set(range(8))


@pytest.mark.parametrize("world,nc", [(8, (4, 3, 5)), (4, (6, 2, 3)), (2, (2, 4, 4)), (8, (1, 2, 2)), (2, (1, 3, 3))])
def test_shell_and_interior_tile_the_block(world, nc):
    from exahype_amd.solvers import CartesianPartition
    part = CartesianPartition(world, 0, 3)
    shell, interior = part.shell_and_interior(nc)
    count = np.zeros(nc, dtype=int)
    for lo, hi in shell + [interior]:
        if all(h > l for l, h in zip(lo, hi)):
            count[tuple(slice(l, h) for l, h in zip(lo, hi))] += 1
    assert np.all(count == 1)
    # every cell touching a partitioned face is in the shell
    mask = np.zeros(nc, dtype=bool)
    for lo, hi in shell:
        mask[tuple(slice(l, h) for l, h in zip(lo, hi))] = True
    for d in range(3):
        if part.partitioned(d):
            assert mask.take(0, axis=d).all() and mask.take(nc[d] - 1, axis=d).all()


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd.solvers import CartesianPartition, HaloExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dim, nc, ts = 3, (3, 2, 4), 10
pdims = %(pdims)r
part = CartesianPartition(world, rank, dim, pdims)
# a global "trace" field whose value encodes (d, side, global cell, entry): every rank can predict its ghosts
G = [nc[a] * part.pdims[a] for a in range(3)]
def global_trace():
    t = torch.zeros((dim, 2) + tuple(G) + (ts,), dtype=torch.float64)
    idx = torch.arange(t.numel(), dtype=torch.float64).reshape(t.shape)
    return idx * 0.5 + 1.0
full = global_trace()
sl = tuple(slice(part.coords[a] * nc[a], (part.coords[a] + 1) * nc[a]) for a in range(3))
local = full[(slice(None), slice(None)) + sl].contiguous()
hx = HaloExchange(part, nc, ts, torch.device("cpu"))
hx.pack(local); hx.start(); hx.finish()
for d in range(dim):
    if not part.partitioned(d):
        assert hx.ghost[2 * d] is None and hx.ghost[2 * d + 1] is None
        continue
    lo_layer = (part.coords[d] * nc[d] - 1) %% G[d]            # global layer just below my block: its R traces
    hi_layer = ((part.coords[d] + 1) * nc[d]) %% G[d]           # layer just above: its L traces
    other = tuple(s for a, s in enumerate(sl) if a != d)
    want_lo = full[d, 1].select(d, lo_layer)[other].reshape(-1, ts)
    want_hi = full[d, 0].select(d, hi_layer)[other].reshape(-1, ts)
    assert torch.equal(hx.ghost[2 * d], want_lo), ("low ghost", rank, d)
    assert torch.equal(hx.ghost[2 * d + 1], want_hi), ("high ghost", rank, d)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world,pdims", [(2, [2, 1, 1]), (2, [1, 1, 2]), (4, [2, 2, 1]), (3, [1, 3, 1]), (8, [2, 2, 2])])
def test_halo_exchange_gloo(tmp_path, world, pdims):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, pdims=pdims))
    port = 29500 + (os.getpid() + world * 7 + pdims[2]) % 2000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
