R=$PWD
cd /tmp; export TMPDIR=/tmp
echo "##### error-controlled leg: product vs ec1 (wave-uniform branch)"
for V in "" ec1; do
  if [ -n "$V" ]; then export GEMX_UNIT_DIR=$R/variants/$V; else unset GEMX_UNIT_DIR; fi
  python - <<PY
import sys, time, torch
sys.path.insert(0, "$R")
import gym_electric_motor_amd as ga
n, K = 65536, 1000
env = ga.make("Cont-SC-SCIM-v0", n_envs=n, tau=1e-4, ode_solver=ga.ScipyOdeSolver())
ps = env.physical_system
a = torch.rand((K, n, 3), device="cuda") * 2 - 1
o = torch.empty((K, n, 14), device="cuda"); d = torch.empty((K, n), dtype=torch.uint8, device="cuda")
f = env.bind_rollout(a, o, d)
for _ in range(40): f()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print("variant '$V': %.3f ms per launch, frac %.3f" % (dt * 1e3, n * (K * 69 + 48) / dt / 8e12), ps.last_launch()[:90])
PY
done
unset GEMX_UNIT_DIR
echo "##### SALU of the SCIM <2,2> kernel with / without the rate limiter"
for P in "" 0; do
  rm -rf /tmp/pmc_salu
  if [ -n "$P" ]; then export GEMX_PACE_GBPS=$P; else unset GEMX_PACE_GBPS; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_salu -- python $R/bench.py --no-extras --no-pmc --repeats 1 --workload scim --steps 5 --warmup 2 --settle-ms 0 > /tmp/pmc_salu.log 2>&1
  echo "GEMX_PACE_GBPS='$P'"; python $R/tools/pmc_sum.py /tmp/pmc_salu advance
done
