"""CPU oracle for the ETCH hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``etch_amd``) never does; it fails loudly without its HIP library.
"""
