"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the reference CHGNet hot path.

Nothing under ``chgnet_amd/`` (the product) may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use
it, and only as the checker / reported baseline -- never as the thing measured.
"""
