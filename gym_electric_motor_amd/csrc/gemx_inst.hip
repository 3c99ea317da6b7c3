// gemx_inst.hip -- ONE kernel unit: every stepping kernel of one (system, converter unit, dtype), built as its own shared object
//   hipcc -DGEMX_INST_SYS=<0..7> -DGEMX_INST_CONV=<0..11> -DGEMX_INST_F64=<0|1> -shared gemx_inst.hip -o libgemx_u<S>_<C>_<F>.so
// (gym_electric_motor_amd/build.py builds the 38 units in parallel) and loaded by gemx_create with dlopen (gemx_capi.hip: load_unit): a
// handle maps libgemx.so and the one unit it runs.  A unit links nothing of libgemx.so: its error messages go through the sink
// gemx_unit_init receives.
#include <cstdarg>
#include <cstdio>

#include "gemx_kernels.hpp"

#if !defined(GEMX_INST_SYS) || !defined(GEMX_INST_CONV) || !defined(GEMX_INST_F64)
#error "define GEMX_INST_SYS, GEMX_INST_CONV and GEMX_INST_F64"
#endif

namespace gemx {
#if GEMX_INST_F64
using InstReal = double;
#else
using InstReal = float;
#endif
static void (*g_set_error)(const char *) = nullptr;
int fail(int code, const char *fmt, ...) {  // gemx_common.hpp declares it; in a unit it formats and hands the text to libgemx.so's gemx_last_error()
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_set_error != nullptr) g_set_error(buf);
    return code;
}
}  // namespace gemx

extern "C" {
// GEMX_OK if this unit was compiled against the caller's handle layout and ABI
__attribute__((visibility("default"))) int gemx_unit_init(unsigned long long handle_bytes, int abi, void (*set_error)(const char *)) {
    if (handle_bytes != (unsigned long long)sizeof(gemx_handle) || abi != GEMX_ABI_VERSION) return GEMX_ERR_ARG;
    gemx::g_set_error = set_error;
    return GEMX_OK;
}
__attribute__((visibility("default"))) int gemx_unit_launch(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    return gemx::launch_advance_unit<GEMX_INST_SYS, GEMX_INST_CONV, gemx::InstReal>(h, actions, K, obs, done, obs_every, st);
}
}
