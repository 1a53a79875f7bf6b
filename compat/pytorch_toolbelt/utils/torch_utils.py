from pytorch_toolbelt_amd.utils.torch_utils import *  # noqa: F401,F403
