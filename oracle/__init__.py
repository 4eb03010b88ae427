"""oracle/ -- TEST INFRASTRUCTURE.  CPU restatement of the reference's algorithm for the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this package,
and only as the checker.  Nothing under ``quantv2x_amd/`` imports it.

Pinned against golden vectors captured from the reference (``tests/golden/make_golden.py``) by
``tests/test_oracle_golden.py``.  Parity status: *pinned* for every stage except the voxelizer
(spconv ``Point2VoxelCPU3d`` is an un-vendored third-party dependency of the reference, version not
pinned: contract only).
"""
