"""CPU oracle for the LAV per-frame hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the integer/index work, plain
float32 numpy / torch-CPU functional ops for the floating-point layers, and a C
file for the pillar path) of the reference algorithm that the HIP kernels in
lav_amd/csrc implement.  Every function cites the reference file:line it follows.

train_cpu.py: the train-mode PointPillar front end and the frozen teacher of the train_lidar step through torch CPU ops, so that
bench.py's cpu_baseline leg can time the train_full_v2 step on host cores (the product has HIP kernels only for both).

Who may import this: tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg - as the checker, never as the thing measured or shipped.
Nothing under lav_amd/ imports it; the product path raises if the HIP library
is missing (lav_amd/_lib.py).

How it is pinned: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so tests/golden/make_golden.py runs the reference's OWN
Python modules (imported read-only from /root/reference, CPU, with stand-ins for
the absent torch_scatter and carla packages) on seeded inputs and commits the
outputs under tests/golden/*.npz; tests/test_oracle_golden.py checks this
oracle against every one of them.  Two third-party pieces remain "parity
unpinned" because neither package exists in this environment:
  * torch_scatter 2.0.7 scatter_max / scatter_mean (restated from its published
    CPU semantics in tests/golden/_shims/torch_scatter.py),
  * CARLA 0.9.10.1 Transform.get_matrix / get_inverse_matrix (restated from
    LibCarla geom/Transform.h in tests/golden/_shims/carla.py).
"""
