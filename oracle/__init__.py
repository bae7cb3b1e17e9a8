"""CPU oracle for the linear auditory-attention-decoding hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a NumPy
restatement of the reference algorithms (google/telluride_decoding v2.1.6),
one function per SURVEY.md section-8 row, each citing the reference file:line
it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it -- always as the checker or the timed CPU
baseline, never as the thing shipped.  The product package
(`telluride_decoding_amd`) never imports `oracle`.

Parity pinning: every function here is checked in `tests/test_oracle_golden.py`
against fixtures in `tests/golden/*.npz`, which were produced by running the
reference's own Python (imported through `tests/golden/ref_shim`) in the build
container with `tests/golden/generate_golden.py`.  Three pieces of arithmetic
live in TensorFlow (a third-party dependency absent from /root/reference and
pinned there only as `tensorflow>=2`): the `tf.signal.frame` lag builder, the
Keras `Dense` forward and Keras' metric averaging.  For those the restatement
is pinned by the literal matrices in the reference's tests
(test/brain_data_test.py:291-294, 317-320, 232-272) and otherwise "parity
unpinned" beyond the reference tests' inequality thresholds.
"""
