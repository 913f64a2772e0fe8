"""CPU oracle for the lane-fit hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``lanedetection_end2end_amd/`` may import this package: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / the timed CPU baseline -- never as the thing
shipped.  See ``oracle/README.md`` for how the restatement is pinned.
"""
