/*
 * global_gain_rccl.c -- "host code stays C" for the multi-GPU global-gain path (north star: RCCL over xGMI carrying
 * only the tiny AGC / level state).  ONE plain-C process drives N GPUs of a node through the C-ABI only:
 * channels are sharded into N contiguous ranges (no data-path collective), every rank's instance runs
 *     selenite_rx_global_process_f32_device(S, in, out, blockSize, comm[rank])
 * (= fused chain with its own AGC off -> per-block envelope -> ncclAllReduce(ncclMax) -> gain law and scale, all on
 * the instance's stream) when the host is one process per GPU.  A single thread that owns several GPUs has to issue
 * the collective of all its ranks inside one ncclGroupStart/End, so here the three steps are spelled out:
 *     selenite_rx_global_phase1_device (every rank) ; ncclAllReduce x N in a group ; selenite_rx_global_phase2_device
 * each rank on a stream the host made (selenite_rx_set_stream), no host synchronisation in between.  The result is
 * compared with ONE unsharded instance holding all the channels on device 0 (which runs the one-call form with a
 * NULL communicator): max is exact, so sharded == unsharded bit for bit.
 *
 * RCCL enters only through its public C API (rccl.h: ncclCommInitAll, ncclAllReduce, ncclGroupStart/End).
 * Build: gcc -O2 -I../../include -I/opt/rocm/include global_gain_rccl.c -L.. -lselenite_rx -L/opt/rocm/lib -lrccl ...
 * Usage: global_gain_rccl [ngpus (default: all)] [channels_total] [samples_per_call] [calls]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "selenite_rx.h"

#define MAXDEV 8

/* Watchdog (the first multi-GPU run must not sit in a hung RCCL bootstrap until somebody's outer limit): a thread
 * that, GLOBAL_GAIN_TIMEOUT_S seconds (default 300) after start, names the stage the host has reached and leaves with _exit(3). */
static const char *volatile g_stage = "start";
static volatile int g_done = 0;
static void *watchdog(void *arg)
{
    const long limit = (long)(intptr_t)arg;
    for (long t = 0; t < limit * 10 && !g_done; ++t) usleep(100000);
    if (!g_done) {
        fprintf(stderr, "global_gain_rccl: still at '%s' after %ld s -- giving up (exit 3)\n", g_stage, limit);
        fflush(stderr);
        _exit(3);
    }
    return NULL;
}

static int fail(const char *what) { fprintf(stderr, "global_gain_rccl: %s: %s\n", what, selenite_rx_error_string(NULL)); return 1; }

int main(int argc, char **argv)
{
    {
        const char *e = getenv("GLOBAL_GAIN_TIMEOUT_S");
        const long limit = e && atol(e) > 0 ? atol(e) : 300;
        pthread_t th;
        if (pthread_create(&th, NULL, watchdog, (void *)(intptr_t)limit) == 0) pthread_detach(th);
        if (getenv("GLOBAL_GAIN_SELFTEST_HANG")) { g_stage = "hung on purpose (selftest)"; for (;;) sleep(1); }
    }
    int ndev = selenite_rx_device_count();
    if (ndev < 1) { fprintf(stderr, "global_gain_rccl: no HIP device (there is no CPU fallback)\n"); return 2; }
    if (argc > 1 && atoi(argv[1]) > 0 && atoi(argv[1]) < ndev) ndev = atoi(argv[1]);
    if (ndev > MAXDEV) ndev = MAXDEV;
    const uint32_t total = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    const uint32_t bs = argc > 3 ? (uint32_t)atoi(argv[3]) : 2048u;
    const int calls = argc > 4 ? atoi(argv[4]) : 3;
    const uint32_t nout = bs / 4;
    static float dec[256], hilb[63], dly[63];
    selenite_rx_config cfg;
    selenite_rx_instance *rx[MAXDEV] = { 0 }, *whole = NULL;
    float *d_in[MAXDEV], *d_out[MAXDEV], *d_env[MAXDEV], *w_in, *w_out;
    hipStream_t stream[MAXDEV];
    uint32_t first[MAXDEV + 1];
    ncclComm_t comm[MAXDEV];
    int devs[MAXDEV];

    memset(&cfg, 0, sizeof cfg);
    if (selenite_rx_design_lowpass(dec, 256, 0.4 / 4) || selenite_rx_design_hilbert(hilb, dly, 63)) return 1;
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = SELENITE_RX_ABI_VERSION;
    cfg.block = 256; cfg.decim = 4; cfg.nd_taps = 256; cfg.nh_taps = 63;
    cfg.arith = SELENITE_ARITH_CMSIS; cfg.mode = SELENITE_MODE_USB;
    cfg.nco_enable = 1; cfg.nco_step_all = 0x01000000u; cfg.agc_enable = 1; cfg.agc_global = 1;
    cfg.dec_coeffs = dec; cfg.hilb_coeffs = hilb; cfg.delay_coeffs = dly;
    cfg.agc_target = 0.5f; cfg.agc_attack = 0.5f; cfg.agc_decay = 0.05f;
    cfg.agc_gain_min = 1e-3f; cfg.agc_gain_max = 1e4f; cfg.agc_env_floor = 1e-6f; cfg.agc_gain_init = 1.0f;

    for (int r = 0; r <= ndev; ++r) first[r] = (uint32_t)((uint64_t)total * r / ndev);   /* contiguous channel ranges */
    for (int r = 0; r < ndev; ++r) devs[r] = r;
    g_stage = "ncclCommInitAll";
    if (ncclCommInitAll(comm, ndev, devs) != ncclSuccess) { fprintf(stderr, "global_gain_rccl: ncclCommInitAll failed\n"); return 1; }

    /* the communicators as RCCL counts them, and one distinct GPU per rank, before any work is queued */
    for (int r = 0; r < ndev; ++r) {
        int nr = -1;
        if (ncclCommCount(comm[r], &nr) != ncclSuccess || nr != ndev) { fprintf(stderr, "global_gain_rccl: communicator of rank %d counts %d ranks, expected %d\n", r, nr, ndev); return 4; }
        char a[32] = "", b[32] = "";
        (void)selenite_rx_device_pci_bus_id(r, a, sizeof a);
        for (int q = 0; q < r; ++q) {
            (void)selenite_rx_device_pci_bus_id(q, b, sizeof b);
            if (strcmp(a, b) == 0) { fprintf(stderr, "global_gain_rccl: ranks %d and %d share device %s\n", q, r, a); return 4; }
        }
    }
    g_stage = "instances";
    for (int r = 0; r < ndev; ++r) {
        const uint32_t n = first[r + 1] - first[r];
        if (selenite_rx_set_device(r)) return fail("set_device");
        cfg.channels = n;
        if (selenite_rx_init(&rx[r], &cfg)) return fail("init");
        d_in[r] = (float *)selenite_rx_device_alloc((size_t)n * bs * 2 * sizeof(float));
        d_out[r] = (float *)selenite_rx_device_alloc((size_t)n * nout * sizeof(float));
        d_env[r] = (float *)selenite_rx_device_alloc((size_t)(bs / 256) * sizeof(float));
        if (!d_in[r] || !d_out[r] || !d_env[r]) return fail("device_alloc");
        if (hipStreamCreateWithFlags(&stream[r], hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
        if (selenite_rx_set_stream(rx[r], stream[r])) return fail("set_stream");
    }
    if (selenite_rx_set_device(0)) return fail("set_device");
    cfg.channels = total;
    if (selenite_rx_init(&whole, &cfg)) return fail("init (unsharded)");
    w_in = (float *)selenite_rx_device_alloc((size_t)total * bs * 2 * sizeof(float));
    w_out = (float *)selenite_rx_device_alloc((size_t)total * nout * sizeof(float));
    float *h_ref = (float *)malloc((size_t)total * nout * sizeof(float));
    float *h_got = (float *)malloc((size_t)total * nout * sizeof(float));
    if (!w_in || !w_out || !h_ref || !h_got) return fail("alloc");

    int bad = 0;
    g_stage = "process calls (phase 1, grouped ncclAllReduce, phase 2)";
    for (int call = 0; call < calls && !bad; ++call) {
        /* the reference: all channels in one instance */
        if (selenite_rx_set_device(0)) return fail("set_device");
        if (selenite_rx_synth_iq_device(whole, w_in, 0, total, (uint64_t)call * bs, bs, 0x5E1E917Eull)) return fail("synth");
        if (selenite_rx_global_process_f32_device(whole, w_in, w_out, bs, NULL)) return fail("unsharded call");
        if (selenite_rx_sync(whole)) return fail("sync");
        if (selenite_rx_memcpy_d2h(h_ref, w_out, (size_t)total * nout * sizeof(float))) return fail("d2h");
        /* the sharded run: one collective per call, issued for all ranks of this process as a group */
        for (int r = 0; r < ndev; ++r) {
            if (selenite_rx_set_device(r)) return fail("set_device");
            if (selenite_rx_synth_iq_device(rx[r], d_in[r], first[r], first[r + 1] - first[r], (uint64_t)call * bs, bs, 0x5E1E917Eull)) return fail("synth");
        }
        for (int r = 0; r < ndev; ++r) {
            if (selenite_rx_set_device(r)) return fail("set_device");
            selenite_rx_global_phase1_device(rx[r], d_in[r], d_out[r], d_env[r], bs);
        }
        ncclGroupStart();
        for (int r = 0; r < ndev; ++r)
            if (ncclAllReduce(d_env[r], d_env[r], bs / 256, ncclFloat, ncclMax, comm[r], stream[r]) != ncclSuccess) { fprintf(stderr, "ncclAllReduce failed\n"); return 1; }
        ncclGroupEnd();
        for (int r = 0; r < ndev; ++r) {
            if (selenite_rx_set_device(r)) return fail("set_device");
            selenite_rx_global_phase2_device(rx[r], d_out[r], d_env[r], bs);
            if (selenite_rx_status(rx[r])) return fail("sharded call");
        }
        for (int r = 0; r < ndev; ++r) {
            if (selenite_rx_set_device(r)) return fail("set_device");
            if (selenite_rx_sync(rx[r])) return fail("sync");
            if (selenite_rx_memcpy_d2h(h_got + (size_t)first[r] * nout, d_out[r], (size_t)(first[r + 1] - first[r]) * nout * sizeof(float))) return fail("d2h");
        }
        if (memcmp(h_ref, h_got, (size_t)total * nout * sizeof(float)) != 0) bad = 1;
    }
    /* one device per rank, by PCI bus id, and what RCCL itself reports for the communicator: lets a reader verify the ranks */
    printf("{\"host\": \"C\", \"ranks\": %d, \"devices\": [", ndev);
    for (int r = 0; r < ndev; ++r) {
        char bus[32] = "unknown";
        int nr = -1, ur = -1;
        (void)selenite_rx_device_pci_bus_id(r, bus, sizeof bus);
        (void)ncclCommCount(comm[r], &nr);
        (void)ncclCommUserRank(comm[r], &ur);
        printf("%s{\"rank\": %d, \"pci_bus_id\": \"%s\", \"nccl_comm_count\": %d, \"nccl_user_rank\": %d, \"channels\": %u}", r ? ", " : "", r, bus, nr, ur,
               first[r + 1] - first[r]);
    }
    printf("], \"channels\": %u, \"samples_per_call\": %u, \"calls\": %d, \"collective\": \"ncclAllReduce(ncclMax), %u floats per call\", "
           "\"collectives_per_call\": 1, \"sharded_equals_unsharded\": %s}\n", total, bs, calls, bs / 256, bad ? "false" : "true");
    g_stage = "teardown";
    for (int r = 0; r < ndev; ++r) { selenite_rx_free(rx[r]); ncclCommDestroy(comm[r]); }
    selenite_rx_free(whole);
    g_done = 1;
    return bad;
}
