"""CPU oracle for the tiled-inference / TTA / loss hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``pytorch_toolbelt_amd`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker -- never as the thing measured
or shipped.

Each function is a numpy restatement of the reference algorithm (pure index
arithmetic and float64/float32 elementwise math, no torch op chains) and cites
the reference ``file:line`` it follows (paths relative to the upstream
``pytorch_toolbelt`` checkout).

Pinning: ``oracle/make_golden.py`` imports the *unmodified* reference behind
``oracle/ref_shim`` (a cv2/torchvision stand-in) and writes seeded input/output
vectors to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
oracle function against those vectors and against the reference's own
known-answer tests (tests/test_tta.py:31-108, tests/test_losses.py:37-209).
"""
from . import tiles_oracle, tta_oracle, losses_oracle, edges_oracle, ensembling_oracle, pointwise_oracle, volumes_oracle  # noqa: F401
