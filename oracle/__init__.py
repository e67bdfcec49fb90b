"""CPU oracle for the per-frame hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is imported by the product package
(``vision_semantic_segmentation_amd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and there only as the checker.
"""
