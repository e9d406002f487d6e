"""CPU oracle for the partial-convolution inpainting hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``text_segmentation_image_inpainting_amd``) never imports
this package and fails loudly when its HIP library is missing.

The oracle is a stock-PyTorch *CPU* restatement of the reference algorithm
(functional, keyed by the reference's ``state_dict`` names).  It is pinned
against the reference itself: ``tests/golden/make_golden.py`` imports
``/root/reference`` in the build container, drives it with seeded inputs and a
deterministic weight filler, and commits the inputs/outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks the restatement
against those fixtures (parity pinned by import, since the reference ships no
tests or golden vectors of its own -- SURVEY.md section 4 / 8c).
"""
