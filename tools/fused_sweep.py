"""Times the one-pass advection + diffusion kernel at 512^3 for several z chunk lengths (the fused_zchunk option).
  python tools/fused_sweep.py [grid] [chunk ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd import _lib  # noqa: E402
from sopht_mpi_amd.device import DeviceContext  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunks = sys.argv[2:] or ["auto", "24", "32", "48", "65", "87", "129"]
ctx = DeviceContext((n, n, n), 2, np.float32)
rng = np.random.default_rng(0)
u, w, out, scr = (ctx.field(3) for _ in range(4))
for f in (u, w):
    f[0] = ctx.from_numpy(rng.random(ctx.padded_shape, dtype=np.float32))
for c in chunks:
    _lib.call("smi_ctx_set_option", ctx.handle, b"fused_zchunk", 0 if c == "auto" else int(c))
    for rep in range(2):
        _lib.call("smi_prof_enable", ctx.handle, 1)
        _lib.call("smi_prof_reset", ctx.handle)
        for _ in range(10):
            _lib.call("smi_advection_diffusion_update", ctx.handle, out.ptr, u.ptr, w.ptr, scr.ptr, 0.2, 0.05)
        prof = _lib.prof_collect(ctx)
    for k, v in prof.items():
        print(f"zchunk {c:>5s}  {k:22s} median {np.median(v['samples_ms']):.4f} ms  avg {v['ms'] / v['launches']:.4f} ms", flush=True)
