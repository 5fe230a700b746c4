"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot-path algorithms.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package, and only as
the checker.  The product path (`pytracking_amd/`) never imports it and fails loudly when the HIP
library is missing.
"""
