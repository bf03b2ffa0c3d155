/*
 * bench_rx.c -- the north-star call pattern in plain C: a C host calls the MI355X library through
 * the C-ABI only (include/selenite_rx.h), on BASELINE cfg3 (65 536 channels, 256-tap /4 decimator,
 * 63-tap Hilbert SSB, AGC, 4096 complex samples per channel and call), data resident in HBM.
 * Prints what bench.py prints for the same shape (Msamples/s, algorithmic GB/s); used by
 * tests/test_gpu_bench.py to check that the C path and the Python-driven path agree.
 *
 * Build: gcc -O2 -I../../include bench_rx.c -L.. -lselenite_rx -Wl,-rpath,'$ORIGIN/..' -lm -o bench_rx
 * Usage: bench_rx [channels [samples_per_call [iters [arith 0|1|2]]]]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "selenite_rx.h"

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv)
{
    const uint32_t channels = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536u;
    const uint32_t bs = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096u;
    const uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 400u;
    const uint32_t arith = argc > 4 ? (uint32_t)atoi(argv[4]) : SELENITE_ARITH_SPLIT16;
    static float dec[256], hilb[63], dly[63];
    selenite_rx_config cfg;
    selenite_rx_instance *rx = NULL;
    float ms = 0.0f;

    if (selenite_rx_device_count() < 1) { fprintf(stderr, "bench_rx: no HIP device (there is no CPU fallback)\n"); return 2; }
    memset(&cfg, 0, sizeof cfg);
    if (selenite_rx_design_lowpass(dec, 256, 0.4 / 4) || selenite_rx_design_hilbert(hilb, dly, 63)) return 1;
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = SELENITE_RX_ABI_VERSION;
    cfg.channels = channels; cfg.block = 256; cfg.decim = 4; cfg.nd_taps = 256; cfg.nh_taps = 63;
    cfg.arith = (uint8_t)arith; cfg.mode = SELENITE_MODE_USB;
    cfg.nco_enable = 1; cfg.nco_step_all = 0x01000000u; cfg.agc_enable = 1;
    cfg.dec_coeffs = dec; cfg.hilb_coeffs = hilb; cfg.delay_coeffs = dly;
    cfg.agc_target = 0.5f; cfg.agc_attack = 0.5f; cfg.agc_decay = 0.05f;
    cfg.agc_gain_min = 1e-3f; cfg.agc_gain_max = 1e4f; cfg.agc_env_floor = 1e-6f; cfg.agc_gain_init = 1.0f;
    if (selenite_rx_init(&rx, &cfg) != SELENITE_RX_SUCCESS) { fprintf(stderr, "bench_rx: init failed: %s\n", selenite_rx_error_string(NULL)); return 1; }

    float *d_iq = (float *)selenite_rx_device_alloc((size_t)channels * bs * 2 * sizeof(float));
    float *d_audio = (float *)selenite_rx_device_alloc((size_t)channels * (bs / 4) * sizeof(float));
    if (!d_iq || !d_audio) { fprintf(stderr, "bench_rx: device allocation failed\n"); return 1; }
    if (selenite_rx_synth_iq_device(rx, d_iq, 0, channels, 0, bs, 0x5E1E917Eull)) return 1;

    /* clock spin-up (an idle MI355X needs ~100 ms of load to reach its sustained clocks), then the timed calls */
    const double t0 = now_s();
    while (now_s() - t0 < 0.3)
        if (selenite_rx_time_process_device(rx, d_iq, d_audio, bs, 16, &ms)) return 1;
    if (selenite_rx_time_process_device(rx, d_iq, d_audio, bs, iters, &ms)) return 1;
    if (selenite_rx_status(rx)) { fprintf(stderr, "bench_rx: %s\n", selenite_rx_error_string(rx)); return 1; }

    uint64_t rd = 0;
    const uint64_t bytes = selenite_rx_algorithmic_bytes(&cfg, bs, &rd);
    printf("{\"host\": \"C\", \"kernel\": \"%s\", \"channels\": %u, \"samples_per_call\": %u, \"iters\": %u, "
           "\"ms_per_call\": %.4f, \"msamples_per_s\": %.1f, \"algorithmic_GBps\": %.1f}\n",
           selenite_rx_kernel_name(rx), channels, bs, iters, ms, (double)channels * bs / ms / 1e3, (double)bytes / ms / 1e6);
    selenite_rx_device_free(d_iq);
    selenite_rx_device_free(d_audio);
    selenite_rx_free(rx);
    return 0;
}
