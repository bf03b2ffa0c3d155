#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex I/Q through the full Selenite RX chain on MI355X.

One "step" = one selenite_rx_process_f32_device() call over the whole resident batch
(channels x block_size complex samples per GPU).  Inputs are generated on the device by the
library's synthetic generator and are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps K --warmup W            # single GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: channels shard across ranks with NO data-path collective (independent channels,
SURVEY.md 8e) -> weak scaling, channels per GPU fixed.  The only exchange the chain can have is the
optional global-gain AGC (--global-gain): 4 bytes per DSP block all-reduced with MAX over RCCL.

Rank 0 prints ONE JSON line (see the driver contract in the task statement) carrying `roofline`
and, at N=1, `cpu_baseline` (the oracle -- a bit-exact CPU restatement of the CMSIS-DSP chain --
timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 vector (= f32 MFMA) peak
SEED = 0x5E1E917E

WORKLOADS = {
    # name: (baseline config, channels per GPU, block_size per call)
    "cfg3": ("cfg3", 65536, 4096),     # headline: NCO + 256-tap /4 + 63-tap SSB + AGC
    "cfg2": ("cfg2", 4096, 48000 - 48000 % 256),
    "cfg4": ("cfg4", 65536, 4096),
    "cfg5": ("cfg2", 131072, 1024),    # weak-scaling shape of BASELINE cfg5 (cfg2 chain)
}


def flops_per_sample(spec):
    """SURVEY.md 8d algorithmic flops per complex input sample."""
    f = 0.0
    if spec.nd_taps:
        f += 2.0 * 2.0 * spec.nd_taps / spec.decim
    if spec.nh_taps:
        f += 2.0 * spec.nh_taps / spec.decim
    if spec.nco:
        f += 20.0
    if spec.n_biquad:
        f += 9.0 * spec.n_biquad / spec.decim
    return f


def pmc_traffic(workload, arith_name, kernel, channels, bs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r1/traffic.json:
    2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate --pmc passes, calibration in profiles/r1/README.md).
    Only returned when the run is the profiled shape and kernel."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1", "traffic.json")))
    except (OSError, ValueError):
        return None
    e = t.get(workload + "_" + arith_name) or t.get(workload)
    if not e or (channels, bs) != WORKLOADS[workload][1:]:
        return None
    if kernel.split("<")[0] not in e["kernel"]:
        return None
    return int(e["hbm_bytes"])


def cpu_baseline(name, arith, budget_s=12.0):
    """Oracle (kind 'port') on the host cores, bounded sample of the same workload."""
    import numpy as np
    import rxcommon as rc
    cores = min(os.cpu_count() or 1, 64)
    cfg_name, _, bs = WORKLOADS[name]
    bs = min(bs, 4096)
    channels = 32 * cores
    spec = rc.baseline_spec(cfg_name, channels, arith)
    iq = rc.synth_iq(0, channels, 0, bs, SEED)
    chain = rc.CpuChain(spec, "orc")
    chain.process(iq, nthreads=cores)          # warm-up (page in, threads)
    calls, t0 = 0, time.perf_counter()
    while True:
        chain.process(iq, nthreads=cores)
        calls += 1
        el = time.perf_counter() - t0
        if el >= budget_s or calls >= 1000:
            break
    value = channels * bs * calls / el / 1e6
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(value, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d channels x %d samples x %d calls, %s chain, oracle/rx_oracle.c "
                      "(bit-exact vs CMSIS-DSP 1.5.3), gcc -O2 -ffp-contract=off, %d pthreads; cpu=%s"
                      % (channels, bs, calls, cfg_name, cores, model),
            "seconds": round(el, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed device spin-up before the W warmup steps: an idle MI355X sits at ~150 MHz and needs "
                         "~60-100 ms of load to reach its sustained clocks (tools/clock_ramp.py); a streaming DSP "
                         "service runs in that steady state.  0 disables it.")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (default: workload's)")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--arith", default=os.environ.get("SELENITE_BENCH_ARITH", "split16"), choices=["cmsis", "fma", "split16"],
                    help="split16 (default): decimator as f16 hi/lo split-precision MFMA product, <=1e-5 rel vs CMSIS "
                         "(north-star tolerance; measured 1.6e-6); fma: FIR tap loops fused, bit-exact vs the fmaf "
                         "oracle; cmsis: bit-exact CMSIS-DSP arithmetic")
    ap.add_argument("--global-gain", action="store_true")
    ap.add_argument("--io", default="f32", choices=["f32", "q15"],
                    help="f32: the canonical float I/Q in / float audio out signature (headline); q15: the firmware's "
                         "int16 slot format either side (dsp_if.c:286-289, arm_q15_to_float / arm_float_to_q15 fused in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true", help="profiling runs: skip the cpu_baseline and other-arithmetic legs")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SELENITE_BENCH_SHARE_GPU") == "1":      # test rig: several ranks on one device (gloo only)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    torch = None
    if world > 1 or os.environ.get("SELENITE_BENCH_FORCE_DIST") == "1":   # (the env var lets a 1-GPU box exercise this path)
        # torch FIRST: its bundled libamdhip64 (same soname) then serves libselenite_rx.so too, so
        # the process holds exactly one HIP runtime.
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
    elif args.global_gain:
        import torch

    import numpy as np
    import rxcommon as rc
    import selenite_rx as sr

    arith = {"fma": rc.ARITH_FMA, "cmsis": rc.ARITH_CMSIS, "split16": rc.ARITH_SPLIT16}[args.arith]
    cfg_name, channels, bs = WORKLOADS[args.workload]
    channels = args.channels or channels
    bs = args.block_size or bs
    sr.lib().selenite_rx_set_device(local_rank)
    spec = rc.baseline_spec(cfg_name, channels, arith, agc_global=args.global_gain)
    rx = sr.Rx(spec.config())
    nout = bs // spec.decim

    d_in = sr.DeviceBuffer(channels * bs * 8)
    d_out = sr.DeviceBuffer(channels * nout * 4)
    rx.synth_device(d_in.ptr, rank * channels, channels, 0, bs, SEED)
    rx.sync()
    q15 = args.io == "q15"
    if q15:
        if args.global_gain:
            raise SystemExit("--io q15 with --global-gain is not a bench shape")
        # int16 slots: the same synthetic signal through the reference's float->q15 rule
        # (arm_float_to_q15.c:117: *32768, truncate, saturate), converted on the host chunk by chunk
        d_in16 = sr.DeviceBuffer(channels * bs * 4)
        chunk = max(1, min(channels, (64 << 20) // (bs * 8)))
        for c0 in range(0, channels, chunk):
            n = min(chunk, channels - c0)
            f = np.empty((n, bs, 2), np.float32)
            sr.lib().selenite_rx_memcpy_d2h(f.ctypes.data, d_in.ptr + c0 * bs * 8, f.nbytes)
            q = np.clip(np.trunc(f * np.float32(32768.0)), -32768, 32767).astype(np.int16)
            sr.lib().selenite_rx_memcpy_h2d(d_in16.ptr + c0 * bs * 4, q.ctypes.data, q.nbytes)
        d_in.free()
        d_in = d_in16

    def barrier():
        rx.sync()
        if dist is not None:
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    env_t = None
    if args.global_gain:
        env_t = torch.zeros(bs // spec.block, dtype=torch.float32, device="cuda:%d" % local_rank)
        # kernels go on torch's current stream, so phase1 -> all-reduce -> phase2 are ordered on
        # the device and a step needs no host synchronisation
        rx.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        if args.global_gain:
            rx.global_phase1(d_in.ptr, d_out.ptr, env_t.data_ptr(), bs)
            if dist is not None:
                dist.all_reduce(env_t, op=dist.ReduceOp.MAX)     # RCCL over xGMI: 4 B per DSP block
            rx.global_phase2(d_out.ptr, env_t.data_ptr(), bs)
        elif q15:
            rx.process_q15_device(d_in.ptr, d_out.ptr, bs)
        else:
            rx.process_device(d_in.ptr, d_out.ptr, bs)

    spin_t0 = time.perf_counter()
    spin_calls = 0
    while (time.perf_counter() - spin_t0) * 1e3 < args.spinup_ms:      # clock ramp, untimed (see --spinup-ms)
        for _ in range(16):
            step()
        rx.sync()
        spin_calls += 16
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    if args.global_gain:
        for _ in range(args.steps):
            step()
        ev_ms = None
    else:
        # the K timed steps are issued by the library between two HIP events recorded on the
        # stream the kernels run on; the wall clock brackets the same region
        ev_ms = (rx.time_process_q15 if q15 else rx.time_process)(d_in.ptr, d_out.ptr, bs, args.steps)
    barrier()
    t1 = time.perf_counter()
    rx.check()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=("cuda:%d" % local_rank) if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_samples = float(world) * channels * bs * args.steps
        value = total_samples / elapsed / 1e6
        alg_bytes, rd_bytes = rx.algorithmic_bytes(bs)
        if q15:     # int16 slots: 4 B per complex sample in, 2 B per audio sample out; state bytes unchanged
            alg_bytes -= channels * (4 * bs + 2 * nout)
            rd_bytes -= channels * 4 * bs
        k_ms = ev_ms if ev_ms is not None else ms_per_step
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        fl = flops_per_sample(spec) * channels * bs
        out = {
            "metric": "Msamples/s complex I/Q through full RX chain (whole job)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "spinup_ms": args.spinup_ms, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "io": args.io,
            "config": {"workload": "%s: %d channels/GPU x %d complex samples/call, %s" % (
                           args.workload, channels, bs,
                           {"cfg3": "NCO + 256-tap arm_fir_decimate/4 + 63-tap Hilbert SSB (USB) + AGC",
                            "cfg2": "127-tap Hilbert SSB (USB) + AGC",
                            "cfg4": "CW: NCO + 4-stage DF1 biquad @500 Hz + AGC",
                            "cfg5": "127-tap Hilbert SSB (USB) + AGC, cfg5 weak-scaling shape"}[args.workload]),
                       "arith": {rc.ARITH_CMSIS: "cmsis-exact (mul,add)", rc.ARITH_FMA: "fma (<=1e-5 rel vs CMSIS)",
                                 rc.ARITH_SPLIT16: "split16 (f16 hi/lo x3 MFMA decimator, <=1e-5 rel vs CMSIS)"}[arith],
                       "kernel": rx.kernel_name(), "agc": "global" if args.global_gain else "per-channel",
                       "parallelism": "channels sharded x%d, no data-path collective" % world},
            "per_gpu_msamples_s": round(value / world, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None if q15 else pmc_traffic(args.workload, args.arith, rx.kernel_name(), channels, bs),
                         "algorithmic_bytes_per_launch": alg_bytes, "read_bytes_per_launch": rd_bytes,
                         "read_frac": round(rd_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "launch_ms_hip_events": round(k_ms, 4)},
            "fma_roof": {"achieved": round(fl / (k_ms * 1e-3) / 1e12, 2), "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(fl / (k_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4),
                         "flops_per_sample": flops_per_sample(spec)},
        }
        if world == 1 and not args.global_gain and not args.main_only and not q15:
            # the same workload in the other arithmetic contracts, outside the timed region, for the record
            others = {}
            for nm, ar in (("fma", rc.ARITH_FMA), ("cmsis", rc.ARITH_CMSIS)):
                if ar == arith:
                    continue
                spec_x = rc.baseline_spec(cfg_name, channels, ar)
                rx_x = sr.Rx(spec_x.config())
                rx_x.time_process(d_in.ptr, d_out.ptr, bs, max(2, args.warmup))
                ms_x = rx_x.time_process(d_in.ptr, d_out.ptr, bs, max(3, args.steps // 2))
                others[nm] = {"value": round(channels * bs / (ms_x * 1e-3) / 1e6, 2), "unit": "Msamples/s",
                              "ms_per_step": round(ms_x, 4), "kernel": rx_x.kernel_name()}
                rx_x.close()
            others["note"] = ("fma: bit-exact vs the oracle's fmaf restatement; cmsis: bit-exact (0 ULP) vs CMSIS-DSP 1.5.3 "
                              "arithmetic; split16: tolerance-based, <=1e-5 relative per DSP block")
            out["other_arith_modes"] = others
        if world == 1 and not args.no_cpu_baseline and not args.main_only:
            out["cpu_baseline"] = cpu_baseline(args.workload, rc.ARITH_CMSIS)
        print(json.dumps(out), flush=True)

    rx.close()
    d_in.free()
    d_out.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
