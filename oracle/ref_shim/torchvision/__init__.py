"""TEST-HARNESS ONLY: empty torchvision stand-in (reference utils/bboxes_utils.py:7 imports box_iou)."""
from . import ops  # noqa: F401
