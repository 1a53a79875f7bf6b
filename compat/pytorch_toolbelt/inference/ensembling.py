from pytorch_toolbelt_amd.inference.ensembling import *  # noqa: F401,F403
from pytorch_toolbelt_amd.inference.ensembling import __all__  # noqa: F401
