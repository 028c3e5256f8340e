from .eulerian_grid_ops import *  # noqa: F401,F403
from .immersed_boundary_ops import *  # noqa: F401,F403
