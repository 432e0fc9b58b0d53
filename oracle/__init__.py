"""oracle/ — CPU restatement of UDiffText's denoising hot path.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32, functional (state-dict in, tensors out), each function citing the reference
file:line it follows (paths relative to the reference root, ZYM-PKU/UDiffText).

Pinning: the reference ships no tests or golden vectors for this path, so the oracle
is pinned against outputs of the REFERENCE ITSELF, captured in the build container by importing
/root/reference under dependency stubs (``tests/golden/make_golden.py``) and committed as small fixtures
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every fixture.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package,
and only as the checker / the timed CPU baseline — never from the product path (``udifftext_amd``).
"""
