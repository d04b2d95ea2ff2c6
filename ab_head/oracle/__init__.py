"""CPU oracle for the GAN training inner loop (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a from-scratch PyTorch-CPU (fp32) restatement of
the reference's hot path, used only as the *checker* by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  The product
path (``gan_variant_research_amd``) never imports it and has no CPU fallback.

Parity pin: the restatement is checked against golden vectors produced by
importing the reference itself in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
"""
