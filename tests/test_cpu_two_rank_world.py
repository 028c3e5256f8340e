"""CPU, world size 2, launched exactly as the driver launches bench.py
(python -m torch.distributed.run --nproc-per-node 2): the host-side process group the slab
decomposition runs on.  The workers use the package's socket world, not torch.distributed's
gloo group: torch bundles its own HIP runtime, and loading it next to libsophtmi.so (which
links the system ROCm) aborts the process at exit ("double free or corruption")."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_world():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "_two_rank_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_world_waits_for_a_port_that_is_busy_for_a_moment():
    """A launcher may hand out a rendezvous port next to which somebody's short-lived connection sits: the world's
    listening socket retries its bind for a while instead of failing at once, and bench.py's own launcher looks for a
    port below the range of client ports whose neighbours are free."""
    import socket
    import threading
    import time

    sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
    import bench
    from sopht_mpi_amd import world

    assert bench.WORLD_PORT_OFFSET == world._PORT_OFFSET
    port = bench.free_rendezvous_port(4)
    assert 20000 <= port < 28000
    blocker = socket.socket()
    blocker.bind(("127.0.0.1", port + world._PORT_OFFSET))     # where rank 0 will listen
    blocker.listen(1)
    threading.Timer(1.0, blocker.close).start()
    t0 = time.time()
    w = world.SocketWorld(rank=0, size=1, addr="127.0.0.1", port=port)
    assert time.time() - t0 >= 0.5 and w.size == 1
