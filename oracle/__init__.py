"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference RegTR hot path, used as the parity checker.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``regtr_amd``) never does and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * native ops   (oracle/regtr_oracle.cpp)  pinned against the unmodified reference C++ compiled
    into oracle/_ref/ and against tests/golden/preprocess_*.npz.
  * float path   (oracle/regtr_ref.py)      pinned against the reference's own Python modules
    imported from /root/reference (tests/golden/*.npz, generator oracle/make_golden.py).
"""
