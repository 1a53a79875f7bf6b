from pytorch_toolbelt_amd.inference.tiles_3d import *  # noqa: F401,F403
from pytorch_toolbelt_amd.inference.tiles_3d import __all__  # noqa: F401
