from .immersed_body_forcing_grid import EmptyForcingGrid, ImmersedBodyForcingGrid, PrescribedForcingGrid  # noqa: F401
from .immersed_body_flow_interaction import (  # noqa: F401
    ImmersedBodyFlowInteractionMPI,
    RigidBodyFlowInteractionMPI,
    CosseratRodFlowInteraction,
)
from .forcing_grids import (  # noqa: F401,E402
    CosseratRodElementCentricForcingGrid,
    CosseratRodSurfaceForcingGrid,
    RigidSphereBody,
    SphereForcingGrid,
    StraightRodBody,
)
from .rigid_body_dynamics import RigidBodyIntegrator, rod_mass_and_inertia, sphere_mass_and_inertia  # noqa: F401,E402
