"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms (and, under ``_ref/``, builds of the reference's
own sources) used as the parity checker and as the reported CPU baseline.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import from here; the
product package ``glow-tts_amd/`` never does.
"""
