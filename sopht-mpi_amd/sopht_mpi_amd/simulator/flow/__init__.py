from .flow_simulators_3d import UnboundedFlowSimulator3D  # noqa: F401
