"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch fp32 / numpy) of the reference algorithms on the CogReasoner hot path
(LiamZhao326/CogStream, model/*.py). It exists to CHECK the HIP path: only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import it. Nothing under cogstream_amd/ imports it,
and the product path fails loudly when the HIP extension is missing.

Pinning: the reference ships no tests or golden vectors, so the oracle is pinned against
outputs of the reference itself, run in the build container by tests/golden/make_golden.py (which
imports /root/reference) and committed as fixtures under tests/golden/*.npz; tests/test_oracle_golden.py
checks every oracle function against them.

Each function cites the reference file:line it restates.
"""
