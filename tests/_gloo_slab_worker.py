"""Worker of the world-size-2 gloo test: the z-slab data flow of the flow step (scatter, face
halo exchange, stencils with rank-dependent boundary flags, ghost sum, gather) carried by
torch.distributed's gloo backend on CPU tensors, with the oracle's stencils doing the
arithmetic.  libsophtmi is deliberately not loaded here (torch ships its own HIP runtime)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stencils_3d as st  # noqa: E402
from oracle.decomp import ZSlabs  # noqa: E402

dist.init_process_group("gloo")
rank, size = dist.get_rank(), dist.get_world_size()
g, n = 2, (16, 8, 12)
slabs = ZSlabs(n, size, g)
phys = slabs.phys(rank)
rng = np.random.default_rng(3)          # same seed on every rank: same global arrays
glob = rng.standard_normal((3,) + n)
forcing = rng.standard_normal((3,) + n)
mine = slabs.scatter(glob)[rank]
mine_f = slabs.scatter(forcing)[rank]


def exchange_faces(field):
    """The halo pattern of MPIGhostCommunicator3D for slabs: interior edge planes to the z
    neighbours' ghost planes (interior y, x extent), as CPU tensors over gloo."""
    reqs, recvs = [], []
    lo_send = torch.from_numpy(np.ascontiguousarray(field[..., g : 2 * g, g:-g, g:-g]))
    hi_send = torch.from_numpy(np.ascontiguousarray(field[..., -2 * g : -g, g:-g, g:-g]))
    if rank > 0:
        buf = torch.empty_like(lo_send)
        reqs += [dist.isend(lo_send, rank - 1), dist.irecv(buf, rank - 1)]
        recvs.append(("lo", buf))
    if rank < size - 1:
        buf = torch.empty_like(hi_send)
        reqs += [dist.isend(hi_send, rank + 1), dist.irecv(buf, rank + 1)]
        recvs.append(("hi", buf))
    for r in reqs:
        r.wait()
    for side, buf in recvs:
        if side == "lo":
            field[..., :g, g:-g, g:-g] = buf.numpy()
        else:
            field[..., -g:, g:-g, g:-g] = buf.numpy()


def gather(field):
    parts = [torch.empty(field[..., g:-g, g:-g, g:-g].shape, dtype=torch.float64) for _ in range(size)]
    dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(field[..., g:-g, g:-g, g:-g])))
    return np.concatenate([p.numpy() for p in parts], axis=-3)


# serial answers on the global grid
full = ZSlabs(n, 1, g)
ser = full.scatter(glob)[0]
ser_f = full.scatter(forcing)[0]
want_flux = np.zeros_like(ser)
for k in range(3):
    st.diffusion_flux(want_flux[k], ser[k], 0.2, g)
want_curl = np.zeros_like(ser)
st.curl(want_curl, ser, 0.5, g)
want_upd = ser.copy()
st.update_vorticity_from_velocity_forcing(want_upd, ser_f, 0.25, g)

# the same operators on this rank's slab after the halo exchange
exchange_faces(mine)
exchange_faces(mine_f)
flux = np.zeros_like(mine)
for k in range(3):
    st.diffusion_flux(flux[k], mine[k], 0.2, g, phys)
curl = np.zeros_like(mine)
st.curl(curl, mine, 0.5, g, phys)
upd = mine.copy()
st.update_vorticity_from_velocity_forcing(upd, mine_f, 0.25, g)
inner = (slice(None),) + (slice(g, -g),) * 3
for got, want, what in ((flux, want_flux, "flux"), (curl, want_curl, "curl"), (upd, want_upd, "update")):
    np.testing.assert_allclose(gather(got), want[inner], rtol=0, atol=1e-13, err_msg=what)

# ghost sum: contributions spread into the ghost planes reach the neighbour's interior
spread = np.zeros(slabs.local_shape())
spread[-g:, 4, 5] = rank + 1.0          # into my high ghost planes
spread[:g, 3, 6] = 10.0 * (rank + 1.0)  # into my low ghost planes
hi_out = torch.from_numpy(np.ascontiguousarray(spread[-g:]))
lo_out = torch.from_numpy(np.ascontiguousarray(spread[:g]))
if rank < size - 1:
    dist.send(hi_out, rank + 1)
if rank > 0:
    buf = torch.empty_like(hi_out)
    dist.recv(buf, rank - 1)
    spread[g : 2 * g] += buf.numpy()
if rank > 0:
    dist.send(lo_out, rank - 1)
if rank < size - 1:
    buf = torch.empty_like(lo_out)
    dist.recv(buf, rank + 1)
    spread[-2 * g : -g] += buf.numpy()
ref = [np.zeros(slabs.local_shape()) for _ in range(size)]
for r in range(size):
    ref[r][-g:, 4, 5] = r + 1.0
    ref[r][:g, 3, 6] = 10.0 * (r + 1.0)
slabs.ghost_sum(ref)
for a in (slice(0, g), slice(-g, None)):
    spread[a] = 0
spread[:, :g] = 0
spread[:, -g:] = 0
spread[:, :, :g] = 0
spread[:, :, -g:] = 0
assert np.array_equal(spread, ref[rank])
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank} ok")
