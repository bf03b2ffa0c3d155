// rx_diag.h -- the switches of libselenite_rx.so that are NOT configuration.
//
// 1. Experiment knobs (grid sizes, repairs switched off, a deliberately wrong output parity ...) are environment variables of DIAGNOSTIC
//    builds only: diag_env() is getenv() under -DSRX_DIAG (what tools/variants/build.sh builds its A/B libraries with) and a constant
//    NULL otherwise -- the default build reads no environment variable except SELENITE_RX_HOST_CHUNK_MB (a documented user setting,
//    rx_api.hip), and none of them on a process call.
// 2. Kernel-selection overrides the TESTS need in order to reach every product path (generic kernels only, the NCO flavours, the grid of
//    AUTO's rerun pass) are a C-ABI call, selenite_rx_set_plan_option (include/selenite_rx.h): process-wide words, validated, read by
//    init (the rerun grid: at launch, one relaxed load).  Results never depend on them.
#pragma once
#include <cstdlib>
#include <stdint.h>

namespace srx {

#ifdef SRX_DIAG
inline const char *diag_env(const char *name) { return std::getenv(name); }
#else
constexpr const char *diag_env(const char *) { return nullptr; }
#endif

// value of SELENITE_RX_OPT_* `option` (rx_api.hip)
uint32_t plan_option(int option);

}  // namespace srx
