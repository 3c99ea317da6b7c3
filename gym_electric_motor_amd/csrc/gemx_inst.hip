// gemx_inst.hip -- one instantiation unit of the advance kernel, compiled once per (system, converter, dtype):
//   hipcc -DGEMX_INST_SYS=<0|1|2> -DGEMX_INST_CONV=<0|1|2> -DGEMX_INST_F64=<0|1> -c gemx_inst.hip
// (gym_electric_motor_amd/build.py compiles the ten units in parallel and links them with gemx_capi.hip).
#include "gemx_kernels.hpp"

#if !defined(GEMX_INST_SYS) || !defined(GEMX_INST_CONV) || !defined(GEMX_INST_F64)
#error "define GEMX_INST_SYS, GEMX_INST_CONV and GEMX_INST_F64"
#endif

#define GEMX_CAT_(a, b, c, d) a##b##_##c##_##d
#define GEMX_CAT(a, b, c, d) GEMX_CAT_(a, b, c, d)

namespace gemx {
#if GEMX_INST_F64
using InstReal = double;
#else
using InstReal = float;
#endif
// e.g. gemx::launch_unit_1_1_0 = synchronous motor system, Finite-B6C, fp32
int GEMX_CAT(launch_unit_, GEMX_INST_SYS, GEMX_INST_CONV, GEMX_INST_F64)(gemx_handle *h, const void *actions, int K, void *obs,
                                                                        uint8_t *done, int obs_every, hipStream_t st) {
    return launch_advance_unit<GEMX_INST_SYS, GEMX_INST_CONV, InstReal>(h, actions, K, obs, done, obs_every, st);
}
}  // namespace gemx
