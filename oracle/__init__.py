"""CPU oracle for the CTR hot path of wangruichens/recsys.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recsys_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED (by the reference): the reference repository contains no
tests, fixtures or golden vectors, and its arithmetic lives in TensorFlow 1.x,
which is neither vendored nor installable in this environment (SURVEY.md
section 8c).  The restatement below therefore follows the reference scripts
(cited file:line, paths relative to /root/reference) plus the TF-1.13/1.14
semantics they silently invoke.  It is pinned by the
known-answer tests of SURVEY.md Appendix B (FarmHash Fingerprint64 KATs from
upstream TF's string_to_hash_bucket tests, CRC-32C / TFRecord framing KATs,
bucketize table, TF-Adam first-step values, closed-form tiny cases) and by
fp64 finite-difference gradient checks, all in ``tests/``.
"""
