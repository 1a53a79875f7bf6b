"""Drop-in alias: put ``<repo>/compat`` (and the repo root) on ``sys.path`` and existing code that imports
``pytorch_toolbelt.inference.{tiles,tta,functional}``, ``pytorch_toolbelt.losses`` or the handful of
``pytorch_toolbelt.utils`` helpers of the tiled-inference loop runs on the MI355X-native implementation unchanged.
Only the hot-path surface of the reference is provided."""
from pytorch_toolbelt_amd import __version__  # noqa: F401
