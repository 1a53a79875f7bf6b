"""Segmentation losses on fused HIP reductions (drop-in for the hot-path part of ``pytorch_toolbelt.losses``)."""
from .balanced_bce import *  # noqa: F401,F403
from .bitempered_loss import *  # noqa: F401,F403
from .dice import *  # noqa: F401,F403
from .focal import *  # noqa: F401,F403
from .focal_cosine import *  # noqa: F401,F403
from .functional import *  # noqa: F401,F403
from .fused import *  # noqa: F401,F403
from .jaccard import *  # noqa: F401,F403
from .lovasz import *  # noqa: F401,F403
from .logcosh import *  # noqa: F401,F403
from .quality_focal_loss import *  # noqa: F401,F403
from .soft_bce import *  # noqa: F401,F403
from .soft_ce import *  # noqa: F401,F403
from .soft_f1 import *  # noqa: F401,F403
from .wing_loss import *  # noqa: F401,F403
