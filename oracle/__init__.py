"""CPU oracle for the DaliID Person-ReID hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``daliid_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the product path.

Every function is a plain fp32 PyTorch-CPU / numpy restatement of the reference
arithmetic and cites the reference ``file:line`` it follows (paths relative to
``/root/reference/Person-ReID/``).

Parity pinning (see DESIGN.md "Oracle"):
  * losses / cosine schedule / proxy selection / TransReID forward are PINNED
    against outputs of the reference's own ``losses.py``, ``train_encodersKIT.py``
    (``selectProxiesByTriagulation``), ``vit_pytorch.py`` and ``make_models.py``
    imported in the build container; vectors live in ``tests/golden/*.npz`` and were
    produced by ``tests/golden/make_golden.py``.
  * ``torchvision.models.resnet50`` and ``torchreid.metrics.evaluate_rank`` are
    third-party code absent from ``/root/reference`` and from this image (no
    version pin in the reference: it has no requirements file).  Their published
    algorithms are restated here; the reference holds no tests or fixtures for
    them, so those two boundaries are "parity unpinned" by reference artefacts
    and are pinned instead by hand-computed known-answer cases and brute-force
    twins in ``tests/``.
"""
