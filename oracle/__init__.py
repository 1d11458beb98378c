"""CPU oracle for the embedding + match hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (``deep-insight-face_amd/``) never imports this
package and fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle.distance``  -- PINNED against outputs of the reference's own
    importable modules (``deep_insight_face/evaluation/utility.py``,
    ``deep_insight_face/networks/utils.py``) generated in the build container by
    ``tests/gen_golden.py`` and committed under ``tests/golden/``.
  * ``oracle.evalproto`` -- PINNED the same way (``calculate_accuracy``,
    ``calculate_val_far``, ``calculate_roc``).
  * ``oracle.nets``      -- PARITY UNPINNED.  The CNN arithmetic lives in
    un-vendored third-party code (``tensorflow.keras.applications.ResNet50V2``,
    no pinned version; TensorFlow absent from the container) or is absent from
    the reference altogether (IResNet-100, ArcMargin).  The restatement follows
    the public layer definitions and the in-repo head
    builders (``deep_insight_face/networks/triplet.py:102-141``) and is
    cross-checked against an independent torch-CPU implementation only.
"""
