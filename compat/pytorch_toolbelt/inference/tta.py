from pytorch_toolbelt_amd.inference.tta import *  # noqa: F401,F403
from pytorch_toolbelt_amd.inference.tta import __all__  # noqa: F401
from pytorch_toolbelt_amd.inference.tta import TTAWrapper, _deaugment_averaging, ms_labels_deaugment, split_into_chunks  # noqa: F401,E402
