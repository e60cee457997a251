"""CPU oracle for the Monte-Carlo strapdown-INS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product (the package under
``gnss-ins-sim_amd/``) never imports this package and has no CPU fallback.

Each function restates one piece of the reference (Aceinna/gnss-ins-sim @
2024-12-20) and cites the reference file:line it follows.  The restatement is
pinned against the unmodified reference executed in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
"""
