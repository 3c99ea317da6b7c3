"""The numbers of the parity contract, in ONE place (round 5 verdict: FLUX_FLOOR and SIGN_MARGIN lived only in the GPU test file, where a
change silently moves the contract).  DESIGN.md section 2 states them in prose; tests/test_host_cpu.py::test_parity_contract_constants_match_the_design_document
pins both sides, and tests/test_gpu_parity.py imports them from here."""

TOL_FP32 = 1e-4      # fp32 on the device vs the reference's fp64: relative per column, max|delta| / max(max|x_ref|, REL_FLOOR) on normalised states
REL_FLOOR = 1e-3     # ... the floor of that denominator
TOL_FP64_SAME_INTEGRATOR = 1e-9   # the fp64 build against the oracle with the same integrator: absolute
DONE_MARGIN = 1e-5   # done masks exact except where the reference's constraint margin is below this
FLUX_FLOOR = 0.05    # induction machines' field-oriented columns: weighted by min(1, |psi_r| / (FLUX_FLOOR max|psi_r|)); magnitudes unweighted
SIGN_MARGIN = 2e-5   # dead-time lanes: a divergence may only BEGIN where the oracle's smallest current at a sign decision is below this fraction of the limit
DEAD_TIME_MIN_COVER = 0.5  # ... and the lanes together must still cover this share of the run
