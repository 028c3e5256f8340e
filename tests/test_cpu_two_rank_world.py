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
