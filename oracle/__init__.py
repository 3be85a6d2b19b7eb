"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
The product path (``isaacgymenvs_amd``) never imports it and fails loudly if its HIP library is missing.
"""
