"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the GuideGen sampling hot path.  Nothing in the
product package `jointimagegeneration_amd` may import from here: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only
as the checker, never as the thing measured or shipped.
"""
