"""CPU oracle for the eeyore MCMC hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``eeyore_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference (papamarkou/eeyore v0.0.20) in the build container with
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every
function here (numpy and C) against those vectors and against the
known-answer values of the reference's own tests.
"""
