"""CPU oracle for the NTIRE2022_ESR test_demo.py forward path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package (ntire2022_esr_amd) never
imports this module; its ops raise if the HIP library is missing.

Parity status: PINNED -- every model graph here is checked against outputs of
the real reference (imported in the authoring container by tools/gen_golden.py,
committed under tests/golden/) in tests/test_oracle_golden.py.

  oracle.prims       ctypes wrappers over libesr_oracle.so (plain C, fp64 accumulate)
  oracle.models      numpy graphs of IMDN / RFDN / RLFN_cut / BSRN on those primitives
  oracle.torch_port  the same graphs as the ATen op sequence the reference issues
                     (used as the timed CPU baseline: it IS the reference's CPU path
                     restated, oneDNN kernels and all)
"""
