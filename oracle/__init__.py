"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatements (plain torch fp32 / numpy) of the arithmetic the reference's three hot-path services run through
their third-party model packages.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package — as the checker, never as the product path.  The
product (``vision-sam3-yolo-lameless_amd/lmx``) never imports it and has no CPU fallback.

Pinning status: the reference holds no golden vectors for this path and its
model packages (ultralytics, segment_anything) are not installed, so
  * vit.py / preprocess.py are pinned against the ``transformers`` / ``Pillow`` code installed in the container
    (tests/test_oracle_*.py, fixtures under tests/golden/ produced by tests/golden/make_golden.py);
  * yolo.py / nms.py have no importable reference: **parity unpinned** beyond authored known-answer cases and the
    analytic parameter/FLOP counts of SURVEY.md Appendix A.1.
"""
