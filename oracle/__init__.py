"""CPU oracle for the morna index-build + search hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``morna_amd/`` does.

  morna_ref.py    pure-Python/numpy restatement of morna.py's own functions
                  (small cases; it is what generated tests/golden/)
  morna_oracle.c  the same arithmetic in C for sizes Python cannot loop over
  annoy_oracle.c  restatement of the third-party annoy forest (parity unpinned
                  for N > K: annoy is absent from /root/reference, see header)
  capi.py         ctypes bindings to _build/libmorna_oracle.so
"""
