from .functional import *  # noqa: F401,F403
from .tiles import *  # noqa: F401,F403
from .tiles_3d import *  # noqa: F401,F403
from .ensembling import *
from .tta import *  # noqa: F401,F403
