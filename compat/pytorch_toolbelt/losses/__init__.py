from pytorch_toolbelt_amd.losses import *  # noqa: F401,F403
from pytorch_toolbelt_amd.losses import functional  # noqa: F401
