"""TEST INFRASTRUCTURE: CPU oracle for the MSDeformAttn hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  The product package (``unipercept_amd``) never does.
"""
