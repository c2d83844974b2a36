"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference correlation path.

Nothing under ``raft_amd/`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` ("Oracle").
"""
