"""CPU restatement of the reference algorithms of the constrained-decoding path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``genrec_amd/`` imports this package;
it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / baseline, never as the
product path.

Pinning: every function here is checked against the real reference (imported
from its checkout in the build container) by ``tests/golden/make_golden.py``,
which also writes the committed golden vectors under ``tests/golden/``; the
CPU test-suite re-checks the restatement against those vectors and against the
known answers the reference's own tests and docs hold.  The beam-search
arithmetic lives in the third-party ``transformers`` package (the reference
pins ``>=4.51.0,<=4.57.1``; the goldens were captured from transformers 5.15.0,
the version installed here, which has the same vectorised ``_beam_search``).
"""
