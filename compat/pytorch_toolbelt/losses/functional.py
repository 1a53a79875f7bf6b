from pytorch_toolbelt_amd.losses.functional import *  # noqa: F401,F403
from pytorch_toolbelt_amd.losses.functional import reduced_focal_loss  # noqa: F401
