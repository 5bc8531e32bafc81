"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the UniDefense hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  The product path
(``unidefense_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference from
``/root/reference`` (this container only), runs it on the same seeded inputs and
key-seeded parameters and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
"""
