from .flow_simulators_3d import UnboundedFlowSimulator3D  # noqa: F401
from .flow_simulators_2d import UnboundedFlowSimulator2D  # noqa: F401
