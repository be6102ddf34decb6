"""CPU oracle for the DISTS / A-DISTS hot path -- test infrastructure only.

Importable from tests/, bench.py (cpu_baseline leg) and __graft_entry__.smoke().
The shipped package nerf_qa_amd never imports it.
"""
