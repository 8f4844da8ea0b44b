"""CPU oracle for the learner-side update path (TEST INFRASTRUCTURE ONLY).

This package is a CPU restatement of the algorithms the reference
(ikostrikov/pytorch-a2c-ppo-acktr-gail) executes on its update path.  It is
the *checker* for the HIP product path that lives under
``pytorch-a2c-ppo-acktr-gail_amd/``:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it;
* nothing under ``pytorch-a2c-ppo-acktr-gail_amd/`` imports it, ever -- the
  product path fails loudly when the HIP library is missing.

Pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``tests/golden/make_golden.py``
(which imports ``/root/reference`` behind dependency stubs) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the oracle
against those files on every run.

Every function cites the reference file:line it restates.
"""
