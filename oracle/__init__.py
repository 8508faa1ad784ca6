"""CPU oracle for the mixed-residual hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a from-scratch CPU restatement (numpy fp64 / PyTorch-CPU fp32)
of the algorithm the reference implements in

    models/darcy.py:162-233        (Darcy residual + boundary loss)
    utils/image_gradient.py:24-92  (boundary-corrected Sobel gradients)
    models/codec.py:43-370         (DenseED / Decoder)
    utils/practices.py:6-41        (one-cycle LR)
    train_codec_mixed_residual.py:166-240 (train step, eval metrics)

Rules ("Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it -- as the checker / the timed CPU
    baseline, never as the thing shipped.  ``pde_surrogate_amd`` never imports
    ``oracle`` and has no CPU fallback.
  * parity is PINNED: every function here is checked in ``tests/test_oracle_golden.py``
    against golden vectors under ``tests/golden/`` that ``tools/gen_golden.py``
    produced by importing the real reference from ``/root/reference`` in the
    build container (the reference ships no tests or fixtures of its own).
  * config 5's "validated against utils/fenics.py" is PARITY UNPINNED: FEniCS /
    dolfin is not installable here and the reference stores no solver outputs.
"""
