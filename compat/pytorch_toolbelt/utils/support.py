from pytorch_toolbelt_amd.utils.support import *  # noqa: F401,F403
