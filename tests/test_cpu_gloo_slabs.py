"""CPU, world size 2, torch.distributed gloo backend on CPU tensors: the slab data flow
(halo exchange, boundary flags per rank, ghost sum, gather) against the serial oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gloo_two_rank_slabs():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29544",
           os.path.join(ROOT, "tests", "_gloo_slab_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout
