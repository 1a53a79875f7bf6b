from pytorch_toolbelt_amd.inference.tiles import *  # noqa: F401,F403
from pytorch_toolbelt_amd.inference.tiles import __all__  # noqa: F401
