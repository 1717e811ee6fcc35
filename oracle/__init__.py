"""oracle/ — CPU restatement of the reference path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
from here, and only as the checker / reported baseline — never as the thing shipped or measured.
"""
