"""The sharded ADER-DG step (boundary shell first, face-trace exchange on a second stream, interior
overlapped) on real kernels: 2 or 4 ranks share cuda:0 and exchange over gloo (host-staged); the result
must equal the oracle's step of the whole periodic grid.  (RCCL between ranks needs one GPU per rank: the
driver's 8-GPU run exercises that; what the one-GPU box can do over the real transport is a rank that is its own
periodic neighbour -- the two *_over_rccl_send_recv_to_self tests below.)  cfg 3 runs at its own order
(N = 6: the persistent LDS-resident stage A on shell / interior boxes, the dense stage B with ghost
buffers in one and in two directions), cfg 4 at its own (N = 8: level-streamed stage A + 17^3 FV patches)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _run_ranks(tmp_path, text, world, timeout=600):
    script = tmp_path / "worker.py"
    script.write_text(text)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:                                   # exactly the processes started here
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd import solvers as exa
import oracle
from oracle.dg_operators import operators
from tests.util import euler_dg_state
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dim, N, nc = 3, %(N)d, %(nc)r
pdims = %(pdims)r
part = exa.CartesianPartition(world, rank, dim, pdims)
G = tuple(nc[a] * part.pdims[a] for a in range(3))
u = euler_dg_state(G + (N,) * dim, seed=42)
dx = [1.0 / g for g in G]
dt = 0.02 * min(dx) / (2 * N - 1)
s = exa.AderDgSolver(dim, N, nc, dx=dx, part=part, backend_is_gloo=True, one_kernel_step=%(one)r)
sl = tuple(slice(part.coords[a] * nc[a], (part.coords[a] + 1) * nc[a]) for a in range(3))
s.upload(u[sl])
ref = u.reshape(-1).copy()
for k in range(3):
    s.step(dt * (1.0 - 0.1 * k))
    ref = oracle.aderdg_step(ref, dt * (1.0 - 0.1 * k), dx, operators(N), dim, N, 5, oracle.PDE_EULER, N, G)
    assert (s._pending_dt is not None) == bool(%(one)r)
torch.cuda.synchronize()
got = s.download()
want = ref.reshape(u.shape)[sl]
err = np.max(np.abs(got - want)) / np.max(np.abs(want))
assert err < 1e-10, err
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "rel err", err)
'''


@pytest.mark.parametrize("N,nc,pdims", [(3, (2, 3, 2), [2, 1, 1]), (4, (3, 2, 2), [1, 2, 1]), (3, (1, 2, 3), [2, 1, 1]),
                                         (6, (2, 2, 2), [2, 1, 1]),      # cfg 3's order, one partitioned direction
                                         (6, (2, 2, 2), [2, 2, 1]),      # cfg 3's order, 4 ranks: ghosts in two directions
                                         (6, (3, 2, 2), [1, 2, 2])])     # 4 ranks, the other two directions; interior box non-empty in x
def test_sharded_step_equals_global_oracle(tmp_path, N, nc, pdims):
    world = pdims[0] * pdims[1] * pdims[2]
    _run_ranks(tmp_path, WORKER % dict(root=ROOT, N=N, nc=nc, pdims=pdims, one=False), world)


@pytest.mark.parametrize("nc,pdims", [((2, 2, 2), [2, 1, 1]), ((3, 3, 2), [1, 2, 2])])
def test_sharded_one_kernel_step_equals_global_oracle(tmp_path, nc, pdims):
    """the step as one kernel on shards (exa_dg_corrector_predictor): the shell cells' kernel takes the previous step's ghosts, the new traces
    travel while the interior cells run; 2 and 4 ranks, N = 6, a different dt every step"""
    world = pdims[0] * pdims[1] * pdims[2]
    _run_ranks(tmp_path, WORKER % dict(root=ROOT, N=6, nc=nc, pdims=pdims, one=True), world)


XT_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd import solvers as exa
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators
from tests.test_user_pde import coupled_xt_ncp_system, OracleXtPDE
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dim, N, nc, pdims = 2, 3, (2, 3), [2, 1]
part = exa.CartesianPartition(world, rank, dim, pdims)
G = tuple(nc[a] * part.pdims[a] for a in range(dim))
p = coupled_xt_ncp_system(max_dim=2)
u = 1.0 + 0.3 * np.random.default_rng(5).random(G + (N,) * dim + (3,))
dx = [0.9 / G[0], 1.1 / G[1]]
origin = [0.25, -0.5]
dt = 0.03 * min(dx) / (2 * N - 1)
s = exa.AderDgSolver(dim, N, nc, pde=p.register(), n_vars=3, dx=dx, part=part, backend_is_gloo=True, origin=origin, time=0.4)
sl = tuple(slice(part.coords[a] * nc[a], (part.coords[a] + 1) * nc[a]) for a in range(dim))
s.upload(u[sl])
ref, t = u.copy(), 0.4
for _ in range(3):
    s.step(dt)
    ref = A.step_xt(ref, dt, dx, operators(N), OracleXtPDE(p), t=t, origin=origin)
    t += dt
torch.cuda.synchronize()
err = np.max(np.abs(s.download() - ref[sl])) / np.max(np.abs(ref))
assert err < 1e-10, err
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "rel err", err)
'''


def test_sharded_step_with_position_dependent_terms_and_ncp(tmp_path):
    """two shards of a 2-D grid, term set with node coordinates, level times and a non-conservative product: every shard evaluates the terms at
    ITS coordinates (origin + shard offset), the ncp jump term crosses the shard boundary through the ghost traces"""
    _run_ranks(tmp_path, XT_WORKER % dict(root=ROOT), 2)


RCCL_SELF_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd import solvers as exa
import oracle
from oracle.dg_operators import operators
from tests.util import euler_dg_state
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)            # RCCL, one rank: its own neighbour in every direction
dim, N, nc = 3, %(N)d, %(nc)r
part = exa.CartesianPartition(1, 0, dim, exchange_self=(0, 1, 2))
u = euler_dg_state(tuple(nc) + (N,) * dim, seed=9)
dx = [1.0 / c for c in nc]
dt = 0.02 * min(dx) / (2 * N - 1)
s = exa.AderDgSolver(dim, N, nc, dx=dx, part=part)               # device buffers go to ncclSend / ncclRecv as they are
assert s.halo is not None and all(g is not None and g.is_cuda for g in s.halo.ghost)
s.exchange_events = []
s.upload(u)
plain = exa.AderDgSolver(dim, N, nc, dx=dx)
plain.upload(u)
ref = u.reshape(-1).copy()
for _ in range(2):
    s.step(dt)
    plain.step(dt)
    ref = oracle.aderdg_step(ref, dt, dx, operators(N), dim, N, 5, oracle.PDE_EULER, N, tuple(nc))
torch.cuda.synchronize()
got = s.download()
err = np.max(np.abs(got - ref.reshape(u.shape))) / np.max(np.abs(ref))
assert err < 1e-10, err
assert np.array_equal(got, plain.download())                     # shell + interior + ghost buffers == one periodic block, bit for bit
assert len(s.exchange_events) == 2
dist.barrier(); dist.destroy_process_group()
print("rccl self exchange rel err", err)
'''


def test_sharded_step_over_rccl_send_recv_to_self(tmp_path):
    """The one exchange the box's single GPU allows over the real transport: a process group of ONE rank on the nccl (= RCCL) backend,
    every direction wrapped through HaloExchange (ncclSend / ncclRecv to self inside one group call).  cfg 3's order; the result must equal
    both the oracle and the un-sharded periodic block."""
    _run_ranks(tmp_path, RCCL_SELF_WORKER % dict(root=ROOT, N=6, nc=(3, 2, 2)), 1)


LIM_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd import solvers as exa
import oracle
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators
from oracle.limiter_numpy import limited_step
from tests.util import euler_dg_state
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dim, N, nc = %(dim)d, %(N)d, %(nc)r
pdims = %(pdims)r
part = exa.CartesianPartition(world, rank, dim, pdims)
G = tuple(nc[a] * part.pdims[a] for a in range(dim))
u = euler_dg_state(G + (N,) * dim, seed=7)
dx = [1.0 / G[0]] * dim
dt = 0.02 * dx[0] / (2 * N - 1)
mask = np.random.default_rng(11).random(G) < 0.4
mask[(0,) * dim] = True                                          # a troubled cell in the corner: every face is a block or wrap face
s = exa.AderDgSolver(dim, N, nc, dx=dx, part=part, backend_is_gloo=True)
lim = exa.SubcellLimiter(s)
sl = tuple(slice(part.coords[a] * nc[a], (part.coords[a] + 1) * nc[a]) for a in range(dim))
s.upload(u[sl])
ops = operators(N)
def fv(patch, dt, h):
    return oracle.fv_corrected(patch[None], dt, h, dim, patch.shape[0] - 2, 1, 5, 0, 1, oracle.PDE_EULER)[0]
ref = u.copy()
for _ in range(2):
    n = lim.step(dt, mask[sl])
    assert n == int(mask[sl].sum())
    ref = limited_step(ref, mask, dt, dx, ops, A.Euler(), fv)
torch.cuda.synchronize()
got, want = s.download(), ref[sl]
err = np.max(np.abs(got - want)) / np.max(np.abs(want))
assert err < 1e-10, err
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "rel err", err)
'''


@pytest.mark.parametrize("dim,N,nc,pdims", [(2, 3, (2, 3), [2, 1]), (3, 3, (2, 2, 1), [1, 2, 1]), (2, 4, (3, 1), [1, 2]),
                                             (3, 8, (2, 1, 2), [2, 1, 1])])      # cfg 4's order: p = 7, 17^3 patches
def test_sharded_limited_step_equals_global_oracle(tmp_path, dim, N, nc, pdims):
    """cfg 4 across ranks: troubled cells at a block face take their FV halo from the neighbour block's subcell layer."""
    _run_ranks(tmp_path, LIM_WORKER % dict(root=ROOT, dim=dim, N=N, nc=nc, pdims=pdims), 2)


RCCL_SELF_LIM_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from exahype_amd import solvers as exa
import oracle
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators
from oracle.limiter_numpy import limited_step
from tests.util import euler_dg_state
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dim, N, nc = 3, %(N)d, %(nc)r
part = exa.CartesianPartition(1, 0, dim, exchange_self=(0, 1, 2))
u = euler_dg_state(tuple(nc) + (N,) * dim, seed=7)
dx = [1.0 / nc[0]] * dim
dt = 0.02 * dx[0] / (2 * N - 1)
mask = np.random.default_rng(11).random(nc) < 0.4
mask[(0,) * dim] = True
s = exa.AderDgSolver(dim, N, nc, dx=dx, part=part)
lim = exa.SubcellLimiter(s)
assert lim.hx_layer is not None and not lim.hx_layer.stage      # subcell layers travel as device buffers
s.upload(u)
ops = operators(N)
def fv(patch, dt, h):
    return oracle.fv_corrected(patch[None], dt, h, dim, patch.shape[0] - 2, 1, 5, 0, 1, oracle.PDE_EULER)[0]
ref = u.copy()
for _ in range(2):
    n = lim.step(dt, mask)
    assert n == int(mask.sum())
    ref = limited_step(ref, mask, dt, dx, ops, A.Euler(), fv)
torch.cuda.synchronize()
err = np.max(np.abs(s.download() - ref)) / np.max(np.abs(ref))
assert err < 1e-10, err
dist.barrier(); dist.destroy_process_group()
print("rccl self exchange, limited step: rel err", err)
'''


def test_sharded_limited_step_over_rccl_send_recv_to_self(tmp_path):
    """cfg 4's order (p = 7, 17^3 FV patches) with both limiter exchanges (troubled flags, subcell layers) and the trace exchange over RCCL
    send/recv to self."""
    _run_ranks(tmp_path, RCCL_SELF_LIM_WORKER % dict(root=ROOT, N=8, nc=(2, 1, 2)), 1)


@pytest.mark.parametrize("N,nc", [(3, (2, 2, 2)), (6, (2, 1, 2))])
def test_full_2x2x2_layout_eight_shards_in_one_process(N, nc):
    """configs[3]'s process grid at kernel level: EIGHT shards of a periodic grid -- all three directions partitioned, every shard with
    ghosts on all six faces -- each with its own plan, boundary shell, packed faces (exa_dg_pack_face), ghost buffers and stage B on
    ghosts; the exchange itself is a device copy between the shards' buffers (eight processes cannot share the box's one GPU: at most six
    may, and RCCL needs a GPU per rank; the transport is covered by the gloo runs above, by tests/test_partition_halo.py with world 8
    and by the RCCL-to-self test).  The result must equal the oracle's step of the whole 2 nc grid."""
    import numpy as np
    import torch
    from exahype_amd import solvers as exa
    import oracle
    from oracle.dg_operators import operators
    from tests.util import euler_dg_state
    dim = 3
    G = tuple(2 * c for c in nc)
    u = euler_dg_state(G + (N,) * dim, seed=88)
    dx = [1.0 / g for g in G]
    dt = 0.02 * min(dx) / (2 * N - 1)
    parts = [exa.CartesianPartition(8, r, dim) for r in range(8)]
    assert parts[0].pdims == [2, 2, 2]
    shards = [exa.AderDgSolver(dim, N, nc, dx=dx, part=p, backend_is_gloo=True) for p in parts]
    sl = [tuple(slice(p.coords[a] * nc[a], (p.coords[a] + 1) * nc[a]) for a in range(3)) for p in parts]
    for s, w in zip(shards, sl):
        s.upload(u[w])
    ref = u.reshape(-1).copy()
    for _ in range(2):
        for s in shards:                                          # stage A on the boundary shell, faces packed
            for lo, hi in s.shell:
                s.predictor_volume(dt, lo, hi)
            s._pack_faces()
        for r, s in enumerate(shards):                            # "exchange": my ghost pair of direction d = the peer's send pair
            for d in range(3):
                peer = parts[r].neighbour(d, +1)
                assert peer == parts[r].neighbour(d, -1) and peer != r
                s.halo._ghost_pair[d].copy_(shards[peer].halo._send_pair[d])
        for s in shards:                                          # interior cells, then Riemann + corrector on the ghosts
            lo, hi = s.interior
            if all(h > l for l, h in zip(lo, hi)):
                s.predictor_volume(dt, lo, hi)
            s.riemann_corrector(dt)
        ref = oracle.aderdg_step(ref, dt, dx, operators(N), dim, N, 5, oracle.PDE_EULER, N, G)
    torch.cuda.synchronize()
    want = ref.reshape(u.shape)
    for s, w in zip(shards, sl):
        err = np.max(np.abs(s.download() - want[w])) / np.max(np.abs(want))
        assert err < 1e-10, err
