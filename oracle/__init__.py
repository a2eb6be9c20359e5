"""CPU oracle for the DiffLinker EGNN sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``difflinker_amd``) never imports this package and fails
loudly when its HIP library is missing.

The oracle is a plain-PyTorch (CPU, fp32 or fp64) restatement of the
reference algorithm, written functionally over a ``state_dict``; every
function cites the reference ``file:line`` it follows.  It is pinned against
outputs of the unmodified reference (imported from ``/root/reference`` in the
build container) through the fixtures in ``tests/golden/`` — see
``tests/golden/make_golden.py`` for the generating script.
"""
