"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (numpy + a small plain-C helper) of the reference's
orthoplane-inference hot path, used only as the checker for the HIP path in
``empanada_amd``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import anything from here; the product
package never does (tests/test_no_oracle_in_product.py enforces it).

Parity status: PINNED.  The restatement is checked against
  * the reference's own test fixtures re-expressed as data
    (tests/test_matcher.py, tests/test_consensus.py, tests/test_array_utils.py), and
  * golden vectors produced by importing the reference itself in the build
    container (oracle/gen_golden.py -> tests/golden/*.npz).
Third-party pieces that are absent from the image and therefore restated from
their published contracts, not run: ``skimage.measure.label`` / ``cc3d``
(8-connected multi-value labelling, raster-order ids) -- "parity unpinned" for
that one function; ``numba`` (identity: it only JIT-compiles
plain Python).

Every function cites the reference file:line (relative to /root/reference) it
follows.
"""
