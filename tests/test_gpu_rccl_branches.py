"""GPU, several ranks on one GPU: the library's RCCL branches (ncclSend / ncclRecv halo and ghost-sum groups, the
grouped all-to-all of the FFT transposes on the communication stream, the pencil redistribution, the transposes of the divided 2D solve,
the all-reduce) against the single-rank oracle.

RCCL refuses two ranks on one device, so these branches cannot run over RCCL itself on a one-GPU box.  Here the
same library objects are linked against a stand-in for librccl (tests/rccl_standin/standin.cpp: the dozen RCCL
entry points carried over Unix sockets between the rank processes, with every message's length checked against
what the receiver expects) and the workers of tests/test_gpu_slabs.py run with SOPHT_MI_TRANSPORT=rccl.  What this
pins: peers, counts, offsets and the send / receive order inside every group.  The stand-in queues its work on the
stream each call names, as RCCL does, and the "held back" cases delay every transfer by some milliseconds: a kernel
that is not ordered behind the exchange that feeds it (the interior / edge split of the one-pass advection +
diffusion kernel, the component pipeline of the Poisson solve on its communication stream) would then read stale
bytes and miss the oracle."""
import os
import subprocess
import sys

import pytest

from test_gpu_slabs import _failure_digest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "rccl_standin", "libsophtmi_standin.so")


def _run(worker, nranks, port, args, delay_ms=0, chunks=1):
    if not os.path.exists(STANDIN):
        pytest.skip("tests/rccl_standin/libsophtmi_standin.so is not built (make standin)")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "rccl"
    env["SOPHT_MPI_AMD_LIB"] = STANDIN
    env["SMI_STANDIN_DELAY_MS"] = str(delay_ms)
    env["SMI_FFT_TRANSPOSE_CHUNKS"] = str(chunks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", worker)] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout
    assert "transport rccl" in out.stdout, "the workers did not run over the RCCL branches"


@pytest.mark.parametrize("nranks,flow_type,mode,grid", [
    (2, "navier_stokes", "fused", "32x16x32"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 nofilter"),   # one exchange of u and w, ghost sum, transposes
    (4, "navier_stokes", "unfused", "32x16x32"),
    (8, "navier_stokes_with_forcing", "fused", "64x32x64"),
    (4, "navier_stokes_with_forcing", "unfused", "64x32x64 pencil"),   # y faces, pencil <-> slab redistribution
    (2, "navier_stokes_with_forcing", "unfused", "32x32x32 yslab"),
    (8, "navier_stokes_with_forcing", "unfused", "64x64x64 pencil"),   # 4 x 2 ranks: the reference's default at 8
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 pencil"),     # the one-call step on pencils
    (8, "navier_stokes_with_forcing", "fused", "64x64x64 nofilter pencil"),
    (4, "navier_stokes_with_forcing", "fused", "64x32x64 yx"),         # x divided: packed x faces, blocks -> slabs
    (8, "navier_stokes_with_forcing", "unfused", "64x32x64 zx"),
    (4, "navier_stokes_with_forcing", "fused", "48x24x40 nofilter"),   # no pruned plan: divided rocFFT transform
])
def test_slab_flow_step_over_the_rccl_branches(nranks, flow_type, mode, grid):
    _run("_slab_worker.py", nranks, 29800 + nranks, [flow_type, mode] + grid.split())


@pytest.mark.parametrize("nranks,args,delay_ms", [(2, "24x20x36", 0), (4, "16x32x64 double", 4),
                                                  (4, "24x24x40 pencil", 4)])
def test_poisson_solver_on_ranks_over_the_rccl_branches(nranks, args, delay_ms):
    """Both back ends of the solver type on slabs and pencils: the all-to-alls of the divided rocFFT transform on the
    compute stream, those of the pruned transform on its communication stream, with transfers held back."""
    _run("_poisson_ranks_worker.py", nranks, 29840 + nranks, args.split(), delay_ms=delay_ms)


@pytest.mark.parametrize("nranks,flow_type,mode,grid", [
    (2, "navier_stokes", "fused", "32x48"),
    (4, "navier_stokes_with_forcing", "fused", "64x48"),
    (4, "navier_stokes_with_forcing", "unfused", "64x48 double"),
])
def test_row_slab_2d_flow_over_the_rccl_branches(nranks, flow_type, mode, grid):
    _run("_rows_worker_2d.py", nranks, 29900 + nranks, [flow_type, mode] + grid.split())


@pytest.mark.parametrize("dim", [3, 2])
def test_general_topology_exchange_over_the_rccl_branches(dim):
    """The axis-by-axis ghost exchange of general decompositions (x divided, periodic) as ncclSend / ncclRecv groups."""
    _run("_ghost_topology_worker.py", 4, 29930 + dim, [str(dim), "quick"])


@pytest.mark.parametrize("nranks,grid,precision", [(2, "16x16x24", "single"), (4, "16x32x24", "double"),
                                                   (8, "32x32x16", "single")])
def test_fft_and_doubled_domain_types_over_the_rccl_branches(nranks, grid, precision):
    """FFTMPI3D and MPIDomainDoublingCommunicator3D across z-slabs: the grouped all-to-all of the transform and the
    ncclSend / ncclRecv pairs of the doubled-domain copies (one send to rank r // 2, two receives on the lower half)."""
    _run("_fft_api_ranks_worker.py", nranks, 29930 + nranks, [grid, precision])


@pytest.mark.parametrize("worker,nranks,args", [
    ("_slab_worker.py", 2, "navier_stokes fused 32x16x32"),
    ("_slab_worker.py", 4, "navier_stokes_with_forcing fused 64x32x64 nofilter"),
    ("_slab_worker.py", 4, "navier_stokes_with_forcing unfused 64x32x64 pencil"),
    ("_slab_worker.py", 4, "navier_stokes_with_forcing fused 64x32x64 pencil"),
    ("_slab_worker.py", 4, "navier_stokes_with_forcing fused 64x32x64 yx"),
    ("_rows_worker_2d.py", 4, "navier_stokes_with_forcing fused 64x48"),
])
def test_transfers_held_back_by_milliseconds(worker, nranks, args):
    _run(worker, nranks, 29950 + nranks, args.split(), delay_ms=4)


@pytest.mark.parametrize("delay_ms", [0, 4])
@pytest.mark.parametrize("nranks,chunks,args", [
    (2, 2, "navier_stokes fused 64x16x32 nofilter"),
    (4, 4, "navier_stokes_with_forcing fused 64x32x64 nofilter"),
    (4, 2, "navier_stokes fused 192x16x32"),
    (8, 4, "navier_stokes_with_forcing fused 64x32x64"),     # slabs of 8 planes: chunks of 2
    (8, 2, "navier_stokes_with_forcing fused 64x32x64"),
])
def test_transposes_in_z_chunks(nranks, chunks, args, delay_ms):
    """Every spectral transpose cut into 2 or 4 chunks of planes (SMI_FFT_TRANSPOSE_CHUNKS): the x transforms of one
    chunk run while another chunk travels on the communication stream.  With every transfer held back by 4 ms, an x
    kernel that is not ordered behind the chunk it reads (or a transfer that starts before its chunk is written)
    misses the oracle."""
    _run("_slab_worker.py", nranks, 29960 + nranks + chunks, args.split(), delay_ms=delay_ms, chunks=chunks)


def _single_and_slabs(worker_name, n, steps, nranks, port, out_dir, extra_env=None, extra_args=()):
    """Runs a worker on one rank (in-tree library) and on `nranks` ranks over the RCCL branches."""
    worker = os.path.join(ROOT, "tests", worker_name)
    env = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SOPHT_MPI_AMD_LIB", "SOPHT_MI_TRANSPORT"):
        env.pop(key, None)
    env.update(extra_env or {})
    single = subprocess.run([sys.executable, worker, str(n), str(steps), out_dir, *extra_args], env=env,
                            capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, _failure_digest(single)
    env["SOPHT_MI_TRANSPORT"] = "rccl"
    env["SOPHT_MPI_AMD_LIB"] = STANDIN
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker, str(n), str(steps), out_dir, *extra_args]
    slabs = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert slabs.returncode == 0, _failure_digest(slabs)
    assert slabs.stdout.count("transport rccl") == nranks


@pytest.mark.parametrize("n,nranks,chunks", [(512, 8, 4), (512, 8, 1), (512, 4, 2), (512, 2, 1)])
def test_benchmark_workload_over_slabs_equals_the_single_rank_run(tmp_path, n, nranks, chunks):
    """The scaling configuration itself - the 512^3 benchmark workload cut into 8 z-slabs of 64 planes, 64 spectral
    columns per rank in rows of 512 bytes - run over the RCCL branches for two steps and compared with the single-rank run
    of the same workload: time step, maximum vorticity, divergence norm, the sum of every interior plane of the
    vorticity and of |velocity|, and - element by element - the two interior planes on either side of every slab
    boundary plus one plane in the middle of a slab (a layout error that only shows at 64 columns per rank, a
    swapped pair of rows inside a plane, would pass the sums)."""
    import json

    import numpy as np

    from helpers import assert_close

    if not os.path.exists(STANDIN):
        pytest.skip("tests/rccl_standin/libsophtmi_standin.so is not built (make standin)")
    out_dir = str(tmp_path)
    nzl = n // nranks
    dump = sorted({b + d for b in range(nzl, n, nzl) for d in (-2, -1, 0, 1)} | {nzl // 2})
    _single_and_slabs("_full_size_ranks_worker.py", n, 2, nranks, 29990 + nranks, out_dir,
                      {"SMI_FFT_TRANSPOSE_CHUNKS": str(chunks)}, [",".join(map(str, dump))])
    for z in dump:
        one = np.load(os.path.join(out_dir, f"plane_1_{z}.npy"))
        many = np.load(os.path.join(out_dir, f"plane_{nranks}_{z}.npy"))
        for k, name in enumerate(("wx", "wy", "wz", "ux", "uy", "uz")):
            assert_close(many[k], one[k], np.float32, factor=1.0, what=f"{name} of plane {z}, {nranks} slabs vs one")
    a = json.load(open(os.path.join(out_dir, "diag_1.json")))
    b = json.load(open(os.path.join(out_dir, f"diag_{nranks}.json")))
    assert abs(a["dt"] - b["dt"]) <= 1e-6 * a["dt"] and abs(a["dt_after"] - b["dt_after"]) <= 1e-5 * a["dt_after"]
    assert abs(a["max_vorticity"] - b["max_vorticity"]) <= 1e-5 * a["max_vorticity"]
    assert abs(a["divergence_l2"] - b["divergence_l2"]) <= 1e-3 * a["divergence_l2"] + 1e-7
    whole = np.load(os.path.join(out_dir, "planes_1_0.npy"))
    parts = np.concatenate([np.load(os.path.join(out_dir, f"planes_{nranks}_{r}.npy")) for r in range(nranks)], axis=1)
    assert whole.shape == parts.shape == (6, n)
    for k in range(6):
        scale = np.abs(whole[k]).max()
        assert np.abs(whole[k] - parts[k]).max() <= 2e-5 * scale, (k, np.abs(whole[k] - parts[k]).max(), scale)


@pytest.mark.parametrize("n,nranks", [(256, 4), (256, 8)])
def test_benchmark_workload_over_pencils_equals_the_single_rank_run(tmp_path, n, nranks):
    """The benchmark workload on the reference's default decomposition (2 x 2 and 4 x 2 pencils over z and y) through
    the one-call step - one-pass advection + diffusion across the inner y faces, Poisson solve routed through the
    slab decomposition - over the RCCL branches, against the single-rank run: diagnostics, plane sums, and element by
    element the planes next to every z boundary and one mid-block plane (whole planes, so the rows on either side
    of every y boundary are among them)."""
    import json

    import numpy as np

    from helpers import assert_close

    if not os.path.exists(STANDIN):
        pytest.skip("tests/rccl_standin/libsophtmi_standin.so is not built (make standin)")
    out_dir = str(tmp_path)
    pz, py = {4: (2, 2), 8: (4, 2)}[nranks]
    nzl = n // pz
    dump = sorted({b + d for b in range(nzl, n, nzl) for d in (-2, -1, 0, 1)} | {nzl // 2})
    _single_and_slabs("_full_size_ranks_worker.py", n, 2, nranks, 29940 + nranks, out_dir, None,
                      [",".join(map(str, dump)), "pencil"])
    for z in dump:
        one = np.load(os.path.join(out_dir, f"plane_1_{z}.npy"))
        many = np.concatenate([np.load(os.path.join(out_dir, f"plane_{nranks}_{z}_y{ry}.npy")) for ry in range(py)],
                              axis=1)
        for k, name in enumerate(("wx", "wy", "wz", "ux", "uy", "uz")):
            assert_close(many[k], one[k], np.float32, factor=1.0, what=f"{name} of plane {z}, {nranks} pencils vs one")
    a = json.load(open(os.path.join(out_dir, "diag_1.json")))
    b = json.load(open(os.path.join(out_dir, f"diag_{nranks}.json")))
    assert abs(a["dt"] - b["dt"]) <= 1e-6 * a["dt"] and abs(a["dt_after"] - b["dt_after"]) <= 1e-5 * a["dt_after"]
    assert abs(a["max_vorticity"] - b["max_vorticity"]) <= 1e-5 * a["max_vorticity"]
    assert abs(a["divergence_l2"] - b["divergence_l2"]) <= 1e-3 * a["divergence_l2"] + 1e-7
    whole = np.load(os.path.join(out_dir, "planes_1_0.npy"))
    parts = np.zeros_like(whole)
    for r in range(nranks):
        cz = int(np.load(os.path.join(out_dir, f"block_{nranks}_{r}.npy"))[0])
        parts[:, cz * nzl:(cz + 1) * nzl] += np.load(os.path.join(out_dir, f"planes_{nranks}_{r}.npy"))
    for k in range(6):
        scale = np.abs(whole[k]).max()
        assert np.abs(whole[k] - parts[k]).max() <= 2e-5 * scale, (k, np.abs(whole[k] - parts[k]).max(), scale)


def test_rod_configuration_over_four_slabs_equals_the_single_rank_run(tmp_path):
    """The shape of the reference's flow-past-rod configuration (384^3, filter of order 1, 5 760 surface nodes on an
    inclined rod) on 4 slabs over the RCCL branches against the single-rank run: the slabs take the host path of the
    forcing grid and the Lagrangian communicators, the single rank forms its nodes on the device."""
    import json

    import numpy as np

    if not os.path.exists(STANDIN):
        pytest.skip("tests/rccl_standin/libsophtmi_standin.so is not built (make standin)")
    n, nranks, out_dir = 384, 4, str(tmp_path)
    _single_and_slabs("_rod_ranks_worker.py", n, 3, nranks, 29985, out_dir, {"SOPHT_MI_ORDERED_SPREAD": "1"})
    a = json.load(open(os.path.join(out_dir, "diag_1.json")))
    b = json.load(open(os.path.join(out_dir, f"diag_{nranks}.json")))
    assert a["lag_nodes"] == b["lag_nodes"] == 5760
    assert abs(a["dt"] - b["dt"]) <= 1e-5 * a["dt"]
    assert abs(a["max_vorticity"] - b["max_vorticity"]) <= 1e-4 * a["max_vorticity"]
    assert abs(a["grid_deviation_l2"] - b["grid_deviation_l2"]) <= 1e-4 * a["grid_deviation_l2"]
    fa, fb = np.array(a["net_force_on_rod"]), np.array(b["net_force_on_rod"])
    assert np.abs(fa - fb).max() <= 1e-4 * np.abs(fa).max(), (fa, fb)
    whole = np.load(os.path.join(out_dir, "planes_1_0.npy"))
    parts = np.concatenate([np.load(os.path.join(out_dir, f"planes_{nranks}_{r}.npy")) for r in range(nranks)], axis=1)
    assert whole.shape == parts.shape == (6, n)
    for k in range(6):
        scale = np.abs(whole[k]).max()
        assert np.abs(whole[k] - parts[k]).max() <= 1e-4 * scale, (k, np.abs(whole[k] - parts[k]).max(), scale)
