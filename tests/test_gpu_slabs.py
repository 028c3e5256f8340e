"""GPU, several ranks: z-slab and pencil decompositions (halo exchange, ghost sum, all-to-all FFT transposes,
Lagrangian ownership) against the single-rank oracle.  The ranks share the test box's one GPU,
so the transport is the host-staged one; the data path through the kernels is the same as
with RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _failure_digest(out):
    """The lines of a failed multi-rank run that name its errors (the rank that failed first is often not the last
    to print), then the tail of both streams."""
    errors = [line for line in (out.stdout + out.stderr).splitlines()
              if "Error" in line and "ChildFailedError" not in line and "ConnectionResetError" not in line]
    return "\n".join(errors[:20]) + "\n----\n" + out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("nranks,flow_type,mode,grid", [
    (2, "navier_stokes", "fused", "32x16x32"),
    # without the vorticity filter: the one-pass advection + diffusion leaves its result in the buffer array and the
    # simulator exchanges the two; slabs of 32 and of 16 planes (several z chunks per slab / a single one)
    (2, "navier_stokes", "fused", "64x16x32 nofilter"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 nofilter"),
    (4, "navier_stokes", "unfused", "32x16x32"),
    (2, "navier_stokes_with_forcing", "fused", "32x16x32"),
    (4, "navier_stokes_with_forcing", "fused", "32x16x32"),
    (4, "navier_stokes", "fused", "192x16x32"),   # a 192-point (radix 8 x 3 x 8) transform across the slabs
    (4, "navier_stokes", "fused", "320x16x32"),   # a 320-point (radix 8 x 5 x 8) transform across the slabs
    (8, "navier_stokes_with_forcing", "fused", "64x32x64"),  # the largest rank count of the scaling runs: 8 of 64 columns per rank
    # pencils over z and y (the reference's default decomposition): y faces, edges, the ghost sum's cascade, node
    # ownership by (z, y) block, the Poisson solve through the slab decomposition
    (4, "navier_stokes", "unfused", "32x32x32 pencil"),
    (4, "navier_stokes_with_forcing", "unfused", "64x32x64 pencil"),
    (4, "navier_stokes_with_forcing", "unfused", "64x32x64 nofilter pencil"),
    (8, "navier_stokes_with_forcing", "unfused", "64x64x64 pencil"),
    (2, "navier_stokes_with_forcing", "unfused", "32x32x32 yslab"),
    # the one-call step on pencils: the one-pass advection + diffusion forms the neighbour's face row across the inner
    # y faces, the Poisson solve is routed through the slab decomposition inside the library
    (4, "navier_stokes", "fused", "32x32x32 pencil"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 pencil"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 nofilter pencil"),
    (8, "navier_stokes_with_forcing", "fused", "64x64x64 pencil"),
    (2, "navier_stokes_with_forcing", "fused", "32x32x32 yslab"),
    # x divided: packed x faces in the exchange and the ghost sum, zero bands on physical x faces only, node ownership
    # by x block, the Poisson solve routed through slabs; 16-byte row vectors where the local row allows (32 / 2 + 4)
    (2, "navier_stokes", "unfused", "32x32x32 xslab"),
    (4, "navier_stokes_with_forcing", "unfused", "64x32x64 xslab"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 yx"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 zx nofilter"),
    (8, "navier_stokes_with_forcing", "fused", "64x32x64 zx"),
    # sizes without a pruned plan (24, 20, 36, 40, 48 points): the rocFFT back end's divided transform - rows where
    # they live, all-to-all to spectral columns, doubled (2nz, 2ny) planes per rank
    (2, "navier_stokes", "fused", "24x20x36"),
    (4, "navier_stokes_with_forcing", "fused", "48x24x40 nofilter"),
    (4, "navier_stokes", "fused", "24x24x40 pencil"),
])
def test_slab_flow_step_matches_oracle(nranks, flow_type, mode, grid):
    _slab_run(nranks, flow_type, mode, grid, chunks=1)


@pytest.mark.parametrize("nranks,chunks,flow_type,mode,grid", [
    (2, 2, "navier_stokes", "fused", "64x16x32 nofilter"),
    (4, 4, "navier_stokes_with_forcing", "fused", "64x32x64 nofilter"),
    (4, 2, "navier_stokes", "fused", "192x16x32"),
    (8, 4, "navier_stokes_with_forcing", "fused", "64x32x64"),
])
def test_slab_flow_step_with_chunked_transposes(nranks, chunks, flow_type, mode, grid):
    """The spectral transposes in 2 or 4 chunks of planes (SMI_FFT_TRANSPOSE_CHUNKS) over the host-staged transport:
    the sliced exchange layout and the chunk-wise x transforms against the single-rank oracle."""
    _slab_run(nranks, flow_type, mode, grid, chunks=chunks)


@pytest.mark.parametrize("nranks,args", [
    (2, "24x20x36"), (2, "24x20x36 double"), (4, "16x32x64"), (4, "16x32x64 double"), (4, "40x12x17"),
    (3, "27x10x12 double"), (4, "24x24x40 pencil"), (4, "32x32x32 pencil double"),
])
def test_poisson_solver_on_ranks_matches_oracle(nranks, args):
    """The solver type by itself on z-slabs and through the pencil route: the divided rocFFT transform on sizes with
    and without pruned plans (odd row lengths and three ranks included), the pruned back end where it has plans."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29670 + nranks),
           os.path.join(ROOT, "tests", "_poisson_ranks_worker.py")] + args.split()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout


def _slab_run(nranks, flow_type, mode, grid, chunks):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    env["SMI_FFT_TRANSPOSE_CHUNKS"] = str(chunks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + nranks),
           os.path.join(ROOT, "tests", "_slab_worker.py"), flow_type, mode] + grid.split()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout


@pytest.mark.parametrize("nranks,flow_type,mode,grid", [
    (2, "navier_stokes", "fused", "32x48"),
    (2, "passive_scalar", "unfused", "32x48"),
    (4, "navier_stokes_with_forcing", "fused", "64x48"),
    (4, "navier_stokes_with_forcing", "unfused", "64x48 double"),
    (8, "navier_stokes_with_forcing", "fused", "64x64"),
])
def test_row_slab_2d_flow_matches_oracle(nranks, flow_type, mode, grid):
    """The 2D path over slabs of rows (the reference's default MPIConstruct2D decomposition) against the single-rank
    oracle on the whole plane."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + nranks),
           os.path.join(ROOT, "tests", "_rows_worker_2d.py"), flow_type, mode] + grid.split()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout


def test_rccl_communicator_initialises(smi):
    """The RCCL transport cannot run across ranks on a one-GPU box (RCCL refuses two ranks on one
    device); what can be checked here is that the library's communicator set-up path (unique id,
    ncclCommInitRank, destruction with the context) works against the RCCL of this image."""
    import ctypes as C

    from sopht_mpi_amd._lib import call, lib

    h = C.c_void_p()
    call("smi_ctx_create", C.byref(h), 0, 0, 16, 16, 16, 2, 0, 1)
    buf = C.create_string_buffer(128)
    call("smi_comm_unique_id", buf)
    call("smi_comm_init", h, buf)
    out = (C.c_double * 1)(3.5)
    call("smi_allreduce_f64", h, out, 1, 0)  # one rank: the value comes back unchanged
    assert out[0] == 3.5
    assert lib.smi_ctx_destroy(h) == 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script starts its two ranks itself (a child
    torch.distributed.run), relays rank 0's JSON line and returns the child's status.  The ranks share the GPU over
    the host-staged transport, so the line is marked invalid as a measurement."""
    import json

    env = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--rehearse-on-host-transport"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["config"]["partition"] == "2 z-slabs"
    assert rec["config"]["transpose_chunks"] == 1 and "invalid" in rec and rec["value"] > 0
    # a rank count the launcher did not start is refused, not silently run on one rank
    env["WORLD_SIZE"] = "1"
    bad = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


@pytest.mark.parametrize("dim,nranks", [(3, 4), (2, 4), (3, 8)])
def test_communicators_on_general_topologies(dim, nranks):
    """Topology, ghost, field and Lagrangian communicators on every rank distribution of the reference's own
    communicator tests (tests/test_utils/test_mpi_utils_3d.py:14-125, test_mpi_utils_2d.py:14-101): x divided,
    periodic domains, faces and - with full_exchange - edges and vertices, both precisions, ghost sizes 1 and 2."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29800 + 10 * dim + nranks),
           os.path.join(ROOT, "tests", "_ghost_topology_worker.py"), str(dim)] + (["quick"] if nranks == 8 else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout
