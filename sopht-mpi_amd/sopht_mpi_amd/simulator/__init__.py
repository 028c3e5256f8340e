from .flow import *  # noqa: F401,F403
from .immersed_body import *  # noqa: F401,F403
