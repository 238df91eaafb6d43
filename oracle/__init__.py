"""TEST INFRASTRUCTURE ONLY — CPU oracle of the Surface-Network SpMM hot path.

`oracle.c_oracle`   ctypes binding of oracle/sn_oracle.c (plain C restatement of the reference kernels).
`oracle.ref_blocks` PyTorch restatement of the reference operator layer on the CPU torch.sparse path
                    ("the repo's own CPU torch.sparse path").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
surfacenetworks_amd never does (tests/test_boundary.py enforces it by scanning the sources).
Parity pinning: see the header of oracle/sn_oracle.c and DESIGN.md §5.
"""
