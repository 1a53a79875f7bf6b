from pytorch_toolbelt_amd.inference.functional import *  # noqa: F401,F403
from pytorch_toolbelt_amd.inference.functional import __all__  # noqa: F401
from pytorch_toolbelt_amd.inference.functional import torch_rot180_transpose, torch_transpose_rot180  # noqa: F401,E402
