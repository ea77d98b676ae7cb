"""oracle/ -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.

CPU restatement of the DeepIPR passport hot path (reference kamwoh/DeepIPR):

  np_passport.py  numpy restatement of the per-layer arithmetic (passport conv
                  -> global pool -> gamma/beta -> affine+ReLU, hinge sign loss,
                  signature-bit parsing) and the analytic backward of each op.
  torch_ref.py    plain-PyTorch (stock ATen CPU ops only) restatement of the
                  blocks, the AlexNet/ResNet passport nets and the V1 / V2-V3
                  train steps, used for full-size parity and as the
                  `cpu_baseline` ("port") leg of bench.py.
  patterns.py     deterministic name-keyed tensor fills shared by the fixture
                  generator (tools/gen_golden.py, which imports the real
                  reference) and the tests.

Pinning: every function cites the reference file:line it follows, and the whole
oracle is checked against outputs of the reference itself (imported from
/root/reference in the build container by tools/gen_golden.py, outputs committed
under tests/golden/*.npz) by tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  deepipr_amd/ never does: the product path is the HIP library
behind include/deepipr_hip.h and raises if it is missing.
"""
