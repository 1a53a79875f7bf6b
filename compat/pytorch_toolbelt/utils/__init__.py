from pytorch_toolbelt_amd.utils import *  # noqa: F401,F403
