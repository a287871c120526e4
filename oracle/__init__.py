"""CPU oracle for the zk-light-client signature-aggregation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, link or execute it, and only as the checker.  The
product path (``zk-light-client-implementation_amd``) never falls back to it.

Each module restates one piece of the reference algorithm and cites the
reference ``file:line`` it follows (paths relative to the reference root).
"""
