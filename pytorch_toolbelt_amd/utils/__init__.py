from .support import *  # noqa: F401,F403
from .torch_utils import *  # noqa: F401,F403
