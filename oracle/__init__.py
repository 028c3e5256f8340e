"""CPU oracle for the sopht-mpi Eulerian flow-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy / scipy.fft restatement of the
reference algorithm (fankiat/sopht-mpi).  It exists so that the HIP kernels can be
checked against an independent CPU statement of the same arithmetic.  Nothing in
the product package may import it: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` do.

Pinning status
--------------
The reference cannot be imported in the build container (its top-level imports
``mpi4py``, ``sopht``, ``pystencils``, ``numba``, ``mpi4py_fft`` are absent and
cannot be installed offline), and it ships no golden vectors.  The restatement is
therefore pinned by the reference's own *known-answer* tests only:

* FFT convention == ``scipy.fft.rfftn`` (tests/test_numeric/.../test_fft_mpi_3d.py:105-115)
* partition of unity of the delta kernel: E->L interpolation of a constant field
  2.0 returns 2.0 (tests/.../test_eulerian_lagrangian_grid_communicator_mpi_3d.py:114-149)
* closed forms of ``compute_stable_timestep``, ``get_max_vorticity`` and the
  divergence norm (tests/test_simulator/test_flow/test_flow_simulators_3d.py:333-500)
* grid-deviation norm (tests/test_simulator/immersed_body/test_immersed_body_interaction_mpi.py:181-188)
* analytic Lamb-Oseen / point-source solutions printed by the examples.

Stencil bodies that live in the un-vendored ``sopht`` package (7-point Laplacian
flux, central-difference curl / divergence, ENO3 flux, elementwise helpers) are
restated from their names, docstrings and call sites; for those "parity unpinned"
applies and each function says so.
"""

from . import stencils_3d, poisson, immersed_boundary, flow_3d, decomp  # noqa: F401
