#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex I/Q through the full Selenite RX chain on MI355X.

One "step" = one selenite_rx_process_f32_device() call over the whole resident batch
(channels x block_size complex samples per GPU).  Inputs are generated on the device by the
library's synthetic generator and are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: when a launcher started this process (torch.distributed.run sets
WORLD_SIZE / RANK / LOCAL_RANK) it is one rank of N; otherwise it starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>`) before touching the
GPU and relays rank 0's JSON line.  A world size that differs from --gpus is an error, never a number.

Multi-GPU: channels shard across ranks with NO data-path collective (independent channels,
SURVEY.md 8e) -> weak scaling, channels per GPU fixed.  The only exchange the chain can have is the
optional global-gain AGC (--global-gain): 4 bytes per DSP block all-reduced with MAX over RCCL.

Rank 0 prints ONE JSON line (see the driver contract in the task statement) carrying `roofline`
and, at N=1, `cpu_baseline` (the CMSIS-DSP chain timed on this box's host cores, single core and all
cores, on a bounded sample of the same workload) and `parity` (the timed arithmetic against that CMSIS chain on a
sample of the workload, per DSP block).  Every side leg (`other_*`) gets its own clock spin-up and >= 100 launches and
reports the MEDIAN of per-launch HIP-event durations, so the line reproduces run to run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 vector (= f32 MFMA) peak
TRAFFIC_JSON = next((p for p in (os.path.join(ROOT, "profiles", r, "traffic.json") for r in ("r6", "r5", "r4", "r3", "r2")) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "r3", "traffic.json"))

WORKLOAD_TEXT = {"cfg3": "NCO + 256-tap arm_fir_decimate/4 + 63-tap Hilbert SSB (USB) + AGC",
                 "cfg2": "127-tap Hilbert SSB (USB) + AGC, DSP block 128 (48 000 samples = 375 blocks)",
                 "cfg2_192": "127-tap Hilbert SSB (USB) + AGC, DSP block 192 (48 000 samples = 250 blocks)",
                 "cfg4": "CW: NCO + 4-stage DF1 biquad @500 Hz + AGC",
                 "cfg5": "127-tap Hilbert SSB (USB) + AGC, cfg5 weak-scaling shape"}


def pmc_traffic(workload, arith_name, kernel, channels, bs, workloads):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this round (profiles/r2/traffic.json:
    2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate --pmc passes, calibration in profiles/r1/README.md).  NOT measured
    in this run (the counters need rocprofv3 around the process): returned only for the profiled shape and kernel,
    and labelled as such in the JSON."""
    try:
        t = json.load(open(TRAFFIC_JSON))
    except (OSError, ValueError):
        return None
    e = t.get(workload + "_" + arith_name) or t.get(workload + "_" + {"auto": "split16"}.get(arith_name, arith_name)) or t.get(workload)
    if not e or (channels, bs) != workloads[workload][1:]:
        return None
    if kernel.split("<")[0] not in e["kernel"]:
        return None
    return int(e["hbm_bytes"])


def cpu_baseline(name, workloads, budget_s=10.0):
    """The CMSIS-DSP chain on the host cores, bounded sample of the same workload: single core, then every core
    the box has (one persistent worker thread per core, each with its own chain instance over its own channels;
    ctypes releases the GIL for the duration of a call).

    kind "reference": oracle/_ref/libcmsis_ref.so -- the reference's own CMSIS-DSP 1.5.3 sources compiled with
    `gcc -O2 -ffp-contract=off -DARM_MATH_CM4` (oracle/Makefile) behind the composition harness oracle/ref_chain.c.
    kind "port": oracle/rx_oracle.c, the plain-loop restatement (bit-exact against the former), when _ref is absent.
    Either way this is test infrastructure used as a yardstick here and nowhere in the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import rxcommon as rc
    kind, which = ("reference", "ref") if rc.ref_available() else ("port", "orc")
    cores = os.cpu_count() or 1
    cfg_name, _, bs = workloads[name]
    bs = min(bs, 4096)
    per = 16                                           # channels per worker and call
    spec = rc.baseline_spec(cfg_name, per, rc.ARITH_CMSIS)

    def measure(nworkers, budget):
        chains = [rc.CpuChain(spec, which) for _ in range(nworkers)]
        iqs = [rc.synth_iq(w * per, per, 0, bs, rc.SEED) for w in range(nworkers)]
        pool = ThreadPoolExecutor(nworkers)
        run = lambda w: chains[w].process(iqs[w])
        list(pool.map(run, range(nworkers)))           # warm-up: page in, spin the threads up
        calls, t0 = 0, time.perf_counter()
        while True:
            list(pool.map(run, range(nworkers)))
            calls += 1
            el = time.perf_counter() - t0
            if el >= budget or calls >= 2000:
                break
        pool.shutdown()
        for c in chains:
            c.close()
        return nworkers * per * bs * calls / el / 1e6, calls, el

    v1, c1, e1 = measure(1, budget_s * 0.3)
    vn, cn, en = measure(cores, budget_s * 0.7)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    what = ("CMSIS-DSP 1.5.3 sources of the reference behind oracle/ref_chain.c" if kind == "reference"
            else "oracle/rx_oracle.c (un-tuned plain-loop restatement, bit-exact vs CMSIS-DSP 1.5.3)")
    return {"value": round(vn, 3), "unit": "Msamples/s", "cores": cores, "kind": kind,
            "sample": "%s chain, %d channels x %d samples per worker and call, %d workers x %d calls in %.1f s; %s; "
                      "gcc -O2 -ffp-contract=off; cpu=%s" % (cfg_name, per, bs, cores, cn, en, what, model),
            "single_core": {"value": round(v1, 3), "unit": "Msamples/s", "cores": 1, "calls": c1, "seconds": round(e1, 2)},
            "seconds": round(e1 + en, 2)}


def parity_check(name, workloads, arith, q15, nch=64, calls=2):
    """The timed arithmetic against the CMSIS chain (oracle/_ref when present, else the pinned restatement) on a sample of the
    workload: first `nch` channels, `calls` calls of the bench call length (4096 at most), per DSP block
    max|gpu - ref| / max|ref|.  Part of the cpu_baseline leg: the oracle is the checker here, never timed
    as the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import rxcommon as rc
    import selenite_rx as sr
    kind, which = ("reference", "ref") if rc.ref_available() else ("port", "orc")
    cfg_name, _, bs = workloads[name]
    bs = min(bs, 4096)
    g = sr.Rx(rc.baseline_spec(cfg_name, nch, arith).config())
    o = rc.CpuChain(rc.baseline_spec(cfg_name, nch, rc.ARITH_CMSIS), which)
    na = g.cfg.block // g.cfg.decim
    worst, worst_abs, blocks = 0.0, 0.0, 0
    worst_margin, saturated = -1e30, 0          # int16 slots: max over blocks of d - (1 + 1e-5 m) LSB (<= 0: inside the bar); blocks that touch full scale
    for call in range(calls):
        iq = rc.synth_iq(0, nch, call * bs, bs, rc.SEED)
        if q15:
            iq16 = np.clip(np.trunc(iq * np.float32(32768.0)), -32768, 32767).astype(np.int16)
            yg, yo = g.process_q15(iq16).astype(np.float64), o.process_q15(iq16).astype(np.float64)
        else:
            yg, yo = g.process(iq).astype(np.float64), o.process(iq).astype(np.float64)
        d = np.abs(yg - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        worst = max(worst, float((d / np.maximum(m, 1e-30)).max()))
        worst_abs = max(worst_abs, float(d.max()))
        blocks += d.size
        if q15:
            # the bar of the int16 slot format (include/selenite_rx.h): the float audio in front of arm_float_to_q15 is within
            # 1e-5 of its block maximum, so the int16 words are within 1 LSB + 1e-5 x that maximum x 32768.  The reference's int16 block maximum
            # IS that maximum (in LSB) unless the block touches full scale, where it is only a lower bound: such blocks are counted, not judged
            sat = m >= 32767.0
            saturated += int(sat.sum())
            if (~sat).any():
                worst_margin = max(worst_margin, float((d - (1.0 + 1e-5 * m))[~sat].max()))
    st = g.guard_stats()
    g.close()
    o.close()
    within = (worst_margin <= 0.0) if q15 else (worst <= 1e-5)
    return {"worst_rel": worst, "bar": "per DSP block |gpu - ref| <= 1 LSB + 1e-5 x block maximum (LSB)" if q15 else 1e-5, "within_bar": bool(within),
            "worst_lsb": int(worst_abs) if q15 else None, "worst_margin_lsb": round(worst_margin, 4) if q15 else None,
            "blocks_at_full_scale": saturated if q15 else None,
            "unit": "max|gpu-ref|/max|ref| per DSP block" + (" (int16 output: worst_lsb / worst_margin_lsb are the figures -- margin = difference minus the bar, <= 0 passes; "
                                                              "one LSB of a block that peaks at 11 000 is 9e-5 in worst_rel)" if q15 else ""),
            "blocks": blocks, "channels": nch, "calls": calls, "against": kind, "kernel": None, "guard_blocks": st["blocks"],
            "rerun_channel_calls": st["rerun_channel_calls"]}


class _View:
    """a window into a DeviceBuffer (side_workload's one-allocation placement); the first view's free() releases the allocation"""
    def __init__(self, buf, offset):
        self.buf, self.ptr, self.owner = buf, buf.ptr + offset, offset == 0

    def free(self):
        if self.owner:
            self.buf.free()


# vector instructions of k_cw_fused<4,2,256> per 128-sample chunk of a wavefront (profiles/r6/sq_cfg4.txt: SQ_INSTS_VALU 1.777e8 / 4096 groups / 32 chunks = 1356)
CW_VALU_PER_CHUNK = 1360


def side_workload(name, q15, arith, spinup_ms, iters=100, rank=0, parity=True, channels_override=0, one_allocation=False):
    """One of the other single-GPU BASELINE configurations on its own resident synthetic batch (the `workloads` block of the default line):
    own instance, own buffers, own clock spin-up, `iters` launches with a HIP event between launches -> MEDIAN per-launch duration
    (SURVEY.md 8d); roofline.frac = SURVEY 8d algorithmic bytes / that / 8 TB/s; `traffic` from the committed PMC passes of the same shape;
    parity.worst_rel on a 64-channel sample against the compiled reference.  Buffers are freed before the next workload."""
    import numpy as np
    import selenite_rx as sr
    from selenite_rx import chain as ch
    cfg_name, channels, bs = ch.WORKLOADS[name]
    channels = channels_override or channels               # (--channels: a scaled run scales these too; `traffic` is then None)
    spec = ch.baseline_spec(cfg_name, channels, arith)
    rx = sr.Rx(spec.config())
    nout = bs // spec.decim
    if one_allocation:
        # input and output inside ONE device allocation (f32 slots only): a second placement of the same buffers (profiles/r5/placement_probe.txt,
        # pattern C: the slower mode of cfg2 in every process of that box; on other boxes either)
        pool = sr.DeviceBuffer(channels * bs * 8 + channels * nout * 4)
        d_in, d_out = _View(pool, 0), _View(pool, channels * bs * 8)
    else:
        d_in = sr.DeviceBuffer(channels * bs * 8)
        d_out = sr.DeviceBuffer(channels * nout * 4)
    rx.synth_device(d_in.ptr, rank * channels, channels, 0, bs, ch.SEED)
    rx.sync()
    note = None
    if q15:
        # int16 slots: the synthetic signal through the reference's float->q15 rule (arm_float_to_q15.c:117), converted on the host for the first
        # `rep` channels and repeated (the conversion of 2 GB through numpy would cost the default run seconds; the kernels' time does not
        # depend on which channel's samples a channel carries)
        rep = min(channels, 8192)
        f = np.empty((rep, bs, 2), np.float32)
        sr.lib().selenite_rx_memcpy_d2h(f.ctypes.data, d_in.ptr, f.nbytes)
        q = np.clip(np.trunc(f * np.float32(32768.0)), -32768, 32767).astype(np.int16)
        d_in.free()
        d_in = sr.DeviceBuffer(channels * bs * 4)
        for c0 in range(0, channels, rep):
            n = min(rep, channels - c0)
            sr.lib().selenite_rx_memcpy_h2d(d_in.ptr + c0 * bs * 4, q.ctypes.data, n * bs * 4)
        note = "int16 input: the first %d channels' signal, repeated" % rep if rep < channels else None
    run = (lambda: rx.process_q15_device(d_in.ptr, d_out.ptr, bs)) if q15 else (lambda: rx.process_device(d_in.ptr, d_out.ptr, bs))
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < spinup_ms:
        for _ in range(16):
            run()
        rx.sync()
    rx.guard_clear()
    ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, iters, q15))
    rx.check()
    guard = rx.guard_stats()
    med = float(ms[len(ms) // 2])
    alg_bytes, rd_bytes = rx.algorithmic_bytes(bs)
    if q15:
        alg_bytes -= channels * (4 * bs + 2 * nout)
        rd_bytes -= channels * 4 * bs
    kernel, nco = rx.kernel_name(), rx.nco_path()
    # the same traffic with no arithmetic, right behind the timed launches (same clocks, same memory temperature): what this box can do for
    # this shape at this moment -- the memory-bound shapes move with it from box to box and with what ran before them (d_out is overwritten)
    rx.time_streaming_roof(d_in.ptr, d_out.ptr, bs, 16, q15)
    roof = float(np.median(rx.time_streaming_roof(d_in.ptr, d_out.ptr, bs, 60, q15)))
    pattern = None
    if kernel.startswith("k_cw_fused"):
        # the systolic CW kernel has a fetch pattern of its own (one wavefront = 16 channel rows at once, 1 KB of a row per load): the same
        # bursts, stores, launch shape and residency with no DSP (k_cw_roof, selenite_rx_time_pattern_roof_device)
        rx.time_pattern_roof(d_in.ptr, d_out.ptr, bs, 16, q15, 0)
        p0 = float(np.median(rx.time_pattern_roof(d_in.ptr, d_out.ptr, bs, 60, q15, 0)))
        p1 = float(np.median(rx.time_pattern_roof(d_in.ptr, d_out.ptr, bs, 60, q15, CW_VALU_PER_CHUNK)))
        pattern = {"kernel": "k_cw_roof", "pattern_roof_ms": round(p0, 4), "frac_of_pattern_roof": round(p0 / med, 4),
                   "pattern_roof_with_arith_ms": round(p1, 4), "frac_of_pattern_roof_with_arith": round(p1 / med, 4),
                   "arith_vector_instructions_per_chunk": CW_VALU_PER_CHUNK,
                   "note": "k_cw_fused's own bursts (16 rows x 1 KB per burst, two bursts in flight), stores, persistent launch shape and residency, no DSP -- "
                           "and the same with as many (dependent) vector instructions between two bursts of a wave as the kernel's biquad steps issue "
                           "(SQ_INSTS_VALU of profiles/r6/sq_cfg4.txt / workgroups / chunks).  The pattern costs nothing against 4 KB runs; what separates "
                           "the kernel from the plain pattern roof is that arithmetic: profiles/r6/cfg4_pattern_roof.md"}
    rx.sync()
    rx.close(); d_in.free(); d_out.free()
    wl = dict(ch.WORKLOADS, **{k + "_q15": v for k, v in ch.WORKLOADS.items()})
    traffic = pmc_traffic(name + ("_q15" if q15 else ""), {sr.ARITH_AUTO: "auto", sr.ARITH_CMSIS: "cmsis", sr.ARITH_FMA: "fma", sr.ARITH_SPLIT16: "split16"}[arith],
                          kernel, channels, bs, wl)
    out = {"workload": "%s%s: %d channels x %d complex samples/call, %s" % (name, " (int16 slots)" if q15 else "", channels, bs, WORKLOAD_TEXT[name]),
           "io": "q15" if q15 else "f32", "kernel": kernel, "nco": nco,
           "value": round(channels * bs / (med * 1e-3) / 1e6, 2), "unit": "Msamples/s",
           "ms_per_step": round(med, 4), "ms_min": round(float(ms[0]), 4), "ms_p90": round(float(ms[int(0.9 * len(ms))]), 4), "launches": int(iters),
           "guard_blocks": guard["blocks"],
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (med * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "read_frac": round(rd_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": alg_bytes, "traffic": traffic,
                        "traffic_source": None if traffic is None else os.path.relpath(TRAFFIC_JSON, ROOT) + " (committed rocprofv3 --pmc passes of this kernel and shape)",
                        "streaming_roof_ms": round(roof, 4), "frac_of_streaming_roof": round(roof / med, 4)}}
    if note:
        out["note"] = note
    if pattern:
        out["roofline"]["pattern_roof"] = pattern
    if name == "cfg5":
        out["mall_note"] = ("per-channel state of this shape: %.0f MB read + as much written per call -- inside the 256 MiB Infinity Cache, whose hits the "
                            "memory-side counters count (MI355X_MICROARCH.md): part of the algorithmic bytes of this figure is served on-die"
                            % (channels * 2 * (spec.nh_taps - 1) * 4 / 1e6))
    if parity:
        pc = parity_check(name, ch.WORKLOADS, arith, q15)
        out["parity"] = {k: pc[k] for k in ("worst_rel", "worst_lsb", "worst_margin_lsb", "blocks_at_full_scale", "bar", "within_bar", "blocks", "channels", "calls", "against", "guard_blocks")}
    return out


def cfg5_scaling_leg(env, rank, world, arith, steps, warmup, spinup_ms, channels_override=0, n1_ms=0.0):
    """N > 1: the shape BASELINE.json's weak-scaling config names -- 131 072 channels per GPU of the cfg2 chain, 1024 samples per call -- timed on
    every rank the way the headline is (barrier + synchronize on both sides, MAX over ranks), beside the default cfg3 leg of the same line."""
    import selenite_rx as sr
    from selenite_rx import chain as ch
    cfg_name, channels, bs = ch.WORKLOADS["cfg5"]
    channels = channels_override or channels             # (--channels: a scaled-down run scales this leg with it, 2 x the headline's count as in BASELINE)
    spec = ch.baseline_spec(cfg_name, channels, arith)
    rx = sr.Rx(spec.config())
    d_in = sr.DeviceBuffer(channels * bs * 8)
    d_out = sr.DeviceBuffer(channels * (bs // spec.decim) * 4)
    rx.synth_device(d_in.ptr, rank * channels, channels, 0, bs, ch.SEED)
    rx.sync()
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < spinup_ms:
        for _ in range(16):
            rx.process_device(d_in.ptr, d_out.ptr, bs)
        rx.sync()
    for _ in range(warmup):
        rx.process_device(d_in.ptr, d_out.ptr, bs)
    rx.sync(); env.barrier()
    t0 = time.perf_counter()
    rx.time_process(d_in.ptr, d_out.ptr, bs, steps)
    rx.sync(); env.barrier()
    t1 = time.perf_counter()
    rx.check()
    elapsed = env.max_over_ranks(t1 - t0)
    alg_bytes, _ = rx.algorithmic_bytes(bs)
    kernel = rx.kernel_name()
    rx.close(); d_in.free(); d_out.free()
    ms = elapsed * 1e3 / steps
    eff = {"weak_scaling_efficiency": round(n1_ms / ms, 4), "n1_ms_per_step": n1_ms} if n1_ms > 0 else {}
    return {**eff, "workload": "cfg5: %d channels/GPU x %d complex samples/call, %s" % (channels, bs, WORKLOAD_TEXT["cfg5"]), "kernel": kernel,
            "value": round(float(world) * channels * bs * steps / elapsed / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "scaling": "weak",
            "steps": steps, "ms_per_step": round(ms, 4), "per_gpu_msamples_s": round(channels * bs * steps / elapsed / 1e6, 2),
            "roofline_frac_per_gpu": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "whole-job aggregate over all ranks, max-over-ranks wall clock; no data-path collective"}


def dist_block(env, devices, collectives_per_step, per_rank_ms, comm_count=None):
    """What lets the driver verify the ranks of an N > 1 line: backend, world, the communicator's size as the backend counted it
    (one all-reduce of ones before anything is timed), one device per rank, collectives per step."""
    return {"backend": env.backend or "none", "world": env.world, "comm_count": comm_count, "devices": devices,
            "collectives_per_step": collectives_per_step, "per_rank_ms_per_step": per_rank_ms}


def check_ranks(env, devices, comm_count, share_gpu):
    """Before anything is timed: the communicator holds WORLD_SIZE ranks and every rank sits on its own GPU.  Every rank holds the
    same gathered values, so every rank takes the same decision: nobody is left waiting in a collective."""
    problems = []
    if comm_count != env.world:
        problems.append("the communicator counts %d rank(s), WORLD_SIZE is %d" % (comm_count, env.world))
    if not share_gpu and len(set(devices)) != len(devices):
        problems.append("ranks share a device: %s" % devices)
    if problems:
        if env.rank == 0:
            sys.stderr.write("bench.py: refusing to time this job -- %s\n" % "; ".join(problems))
        env.close()
        sys.exit(4)


def selftest_launch(env, args):
    """--selftest-launch: the launcher / rendezvous / collective plumbing of the N > 1 path without a GPU
    (CPU tests): no library call, gloo only.  Emits the same `dist` block as a real line."""
    env.init_process_group("gloo", use_gpu=False)
    comm = env.comm_count()
    hang = os.environ.get("SELENITE_SELFTEST_HANG_RANK")
    if hang is not None and int(hang) == env.rank:       # test hook: a rank that never reaches the next collective
        env.stage("hung on purpose (selftest)")
        time.sleep(3600)
    if os.environ.get("SELENITE_SELFTEST_FAIL_RANK") == str(env.rank):      # test hook: a rank that dies
        env.stage("failing on purpose (selftest)")
        os._exit(7)
    env.barrier()
    mine = 0.001 * (env.rank + 1)
    worst = env.max_over_ranks(mine)
    devices = env.gather_objects("cpu:%d" % env.local_rank)
    check_ranks(env, devices, comm, False)
    per_rank = env.gather_objects(round(mine * 1e3, 4))
    if env.rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": args.gpus, "world": env.world, "max_over_ranks_s": worst,
                          "dist": dist_block(env, devices, 1 if args.global_gain else 0, per_rank, comm),
                          "config": {"parallelism": "channels sharded x%d, no data-path collective" % env.world}}), flush=True)
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed device spin-up before the W warmup steps: an idle MI355X sits at ~150 MHz and needs "
                         "~60-100 ms of load to reach its sustained clocks (tools/clock_ramp.py); a streaming DSP "
                         "service runs in that steady state.  0 disables it.")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOAD_TEXT))
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (default: workload's)")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--arith", default=os.environ.get("SELENITE_BENCH_ARITH", "auto"), choices=["cmsis", "fma", "split16", "auto"],
                    help="auto (default): many-tap FIR as f16 hi/lo split-precision MFMA product with a per-pass block exponent; "
                         "channels with a DSP block under the parity guard (envelope more than 12 dB under the input) are recomputed "
                         "in the same call by the bit-exact kernel: <=1e-5 rel vs CMSIS on every block; split16: the same "
                         "without the rerun (guarded blocks only counted); fma: FIR tap loops fused, bit-exact vs the fmaf "
                         "oracle; cmsis: bit-exact CMSIS-DSP arithmetic")
    ap.add_argument("--global-gain", action="store_true")
    ap.add_argument("--nco", default="default", choices=["default", "per_channel", "per_channel_grid", "per_channel_grid_wide", "shared_table"],
                    help="NCO flavour of the TIMED instance (cfg3; the same flavours are side legs of the default run): per_channel = every "
                         "channel its own arbitrary step (arm_sin/cos_f32 per sample in the kernel); per_channel_grid = every channel its "
                         "own step on the fs/256 grid (LO period computed once per channel and call); shared_table = one step off that "
                         "grid for all channels (LO table per call, read from L2)")
    ap.add_argument("--io", default="f32", choices=["f32", "q15"],
                    help="f32: the canonical float I/Q in / float audio out signature (headline); q15: the firmware's "
                         "int16 slot format either side (dsp_if.c:286-289, arm_q15_to_float / arm_float_to_q15 fused in)")
    ap.add_argument("--auto-launches", type=int, default=1, choices=[1, 3],
                    help="--arith auto: 1 (the library's default) = the matrix kernel recomputes a channel it guarded itself where it can (the "
                         "no-decimator shapes: one launch per call), 3 = always k_hist_exact + the rerun pass behind it (selenite_rx_set_auto_launches; same bits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true", help="profiling runs: skip the cpu_baseline and other-mode legs")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--n1-ms", type=float, default=0.0,
                    help="ms_per_step of the SAME command at --gpus 1 (a stored figure): the N > 1 line then carries weak_scaling_efficiency = that / its own "
                         "ms_per_step (the driver computes its own from the per-N values; this is for a human running one N)")
    ap.add_argument("--n1-cfg5-ms", type=float, default=0.0, help="... and of the cfg5 shard (workloads.cfg5.ms_per_step of the N = 1 line) for cfg5_weak_scaling")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from selenite_rx import shard                         # pure Python: nothing below touches the GPU yet
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves, as child processes, before any GPU call
        sys.exit(shard.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    env = shard.RankEnv()
    if env.world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); refusing to report a number\n"
                         % (args.gpus, env.world))
        sys.exit(2)
    if args.selftest_launch:
        return selftest_launch(env, args)
    rank, world = env.rank, env.world
    local_rank = 0 if os.environ.get("SELENITE_BENCH_SHARE_GPU") == "1" else env.local_rank   # test rig: ranks share device 0 (gloo)
    comm = 1
    share_gpu = os.environ.get("SELENITE_BENCH_SHARE_GPU") == "1"
    if world > 1 or os.environ.get("SELENITE_BENCH_FORCE_DIST") == "1":      # (the env var lets a 1-GPU box exercise this path)
        env.local_rank = local_rank
        env.init_process_group(args.dist_backend)
        local_rank = env.local_rank                  # (wrapped when the launcher gave this rank fewer visible devices than LOCAL_RANK + 1)
        comm = env.comm_count()
    elif args.global_gain:
        import torch
        env.torch = torch

    import numpy as np
    import selenite_rx as sr
    from selenite_rx import chain as ch

    arith = {"fma": sr.ARITH_FMA, "cmsis": sr.ARITH_CMSIS, "split16": sr.ARITH_SPLIT16, "auto": sr.ARITH_AUTO}[args.arith]
    cfg_name, channels, bs = ch.WORKLOADS[args.workload]
    channels = args.channels or channels
    bs = args.block_size or bs
    sr.lib().selenite_rx_set_device(local_rank)
    if env.dist is not None:
        # every rank on its own GPU, the communicator complete: checked on every rank BEFORE the timed region
        check_ranks(env, env.gather_objects(sr.device_pci_bus_id(local_rank)), comm, share_gpu)
        env.stage("setting up the instance")
    def nco_steps(kind):
        """per-channel NCO steps of the side legs / --nco: (steps or None, environment for the instance's construction)"""
        if kind == "per_channel":
            return (np.arange(channels, dtype=np.uint64) * 0x9E3779B1 % (1 << 26) + 0x00800000).astype(np.uint32), {}
        if kind == "per_channel_grid":          # steps 0 and fs/256 alternating: different per channel, and the bench signal's tone stays in band
            return ((np.arange(channels, dtype=np.uint64) & 1) << 24).astype(np.uint32), {}
        if kind == "per_channel_grid_wide":     # any multiple of fs/256: the tone leaves the pass band on ~3/4 of the channels (the guard's worst case)
            return ((np.arange(channels, dtype=np.uint64) * 0x9E3779B1 >> 7) % 256 << 24).astype(np.uint32), {}
        if kind == "shared_table":
            return None, {sr.OPT_NO_PERIODIC_LO: 1}
        return None, {}

    def make_rx(ar, kind="default", **kw):
        steps, opts = nco_steps(kind)                    # (opts: kernel-selection overrides, selenite_rx_set_plan_option)
        for k, v in opts.items():
            sr.lib().selenite_rx_set_plan_option(k, v)
        try:
            return sr.Rx(ch.baseline_spec(cfg_name, channels, ar, **(dict(kw, nco_steps=steps) if steps is not None else kw)).config())
        finally:
            for k in opts:
                sr.lib().selenite_rx_set_plan_option(k, 0)

    spec = ch.baseline_spec(cfg_name, channels, arith, agc_global=args.global_gain)
    rx = make_rx(arith, args.nco, agc_global=args.global_gain)
    if args.auto_launches != 1:
        rx.set_auto_launches(args.auto_launches)
    nout = bs // spec.decim

    d_in = sr.DeviceBuffer(channels * bs * 8)
    d_out = sr.DeviceBuffer(channels * nout * 4)
    rx.synth_device(d_in.ptr, rank * channels, channels, 0, bs, ch.SEED)
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

    gg = shard.GlobalGainStepper(rx, env, bs // spec.block, local_rank) if args.global_gain else None

    def sync_all():
        rx.sync()
        if gg is not None:
            gg.synchronize()
        env.barrier()

    def step():
        if gg is not None:
            gg.step(d_in.ptr, d_out.ptr, bs)
        elif q15:
            rx.process_q15_device(d_in.ptr, d_out.ptr, bs)
        else:
            rx.process_device(d_in.ptr, d_out.ptr, bs)

    def spin(fn, sync, ms):                                 # clock ramp, untimed (see --spinup-ms)
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e3 < ms:
            for _ in range(16):
                fn()
            sync()

    def leg(rx_x, iters=None, src=None):
        """One side leg on the resident data: its own spin-up (a third of the headline's), then >= 100 launches with a HIP event
        between launches; the MEDIAN per-launch duration is the figure, mean / min / p90 beside it."""
        src = d_in.ptr if src is None else src
        iters = iters or max(100, args.steps // 4)
        spin(lambda: rx_x.process_q15_device(src, d_out.ptr, bs) if q15 else rx_x.process_device(src, d_out.ptr, bs),
             rx_x.sync, args.spinup_ms / 3.0)
        ms = np.sort(rx_x.time_process_each(src, d_out.ptr, bs, iters, q15))
        med = float(ms[len(ms) // 2])
        return {"value": round(channels * bs / (med * 1e-3) / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(med, 4),
                "ms_mean": round(float(ms.mean()), 4), "ms_min": round(float(ms[0]), 4), "ms_p90": round(float(ms[int(0.9 * len(ms))]), 4),
                "launches": int(iters), "kernel": rx_x.kernel_name(), "nco": rx_x.nco_path()}

    env.stage("spin-up and warm-up")
    spin(step, rx.sync, args.spinup_ms)
    for _ in range(args.warmup):
        step()
    env.stage("timed steps")
    rx.guard_clear()
    sync_all()
    t0 = time.perf_counter()
    if gg is not None:
        for _ in range(args.steps):
            step()
        ev_ms = None
    else:
        # the K timed steps are issued by the library between two HIP events recorded on the
        # stream the kernels run on; the wall clock brackets the same region
        ev_ms = (rx.time_process_q15 if q15 else rx.time_process)(d_in.ptr, d_out.ptr, bs, args.steps)
    sync_all()
    t1 = time.perf_counter()
    rx.check()
    env.stage("after the timed steps")
    guard = rx.guard_stats()
    elapsed = env.max_over_ranks(t1 - t0)
    per_rank_ms = env.gather_objects(round((t1 - t0) * 1e3 / args.steps, 4))
    devices = env.gather_objects(sr.device_pci_bus_id(local_rank))
    # per-launch durations (one HIP event between launches), after the timed region: the median SURVEY.md 8d asks for
    each = None
    roof_ms = None
    if gg is None and rank == 0:
        each = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, min(max(args.steps, 20), 200), q15))
        if not args.main_only:
            # the streaming roof of this call's traffic, measured in the same run at the same clocks: a kernel that moves the
            # algorithmic bytes of the call (input, audio, per-channel state both ways) with the fused kernels' access pattern
            # and no arithmetic (selenite_rx_time_streaming_roof_device; d_out is overwritten -- nothing reads it afterwards)
            spin(lambda: rx.time_streaming_roof(d_in.ptr, d_out.ptr, bs, 8, q15), rx.sync, args.spinup_ms / 3.0)
            roof_ms = np.sort(rx.time_streaming_roof(d_in.ptr, d_out.ptr, bs, 200, q15))
    env.barrier()
    cfg5_line = None
    # (every run that has a process group: N > 1, and the one-rank group of SELENITE_BENCH_FORCE_DIST -- the N > 1 code path on a one-GPU box)
    if (world > 1 or env.dist is not None) and args.workload == "cfg3" and not args.global_gain and not q15 and not args.block_size:
        env.stage("cfg5 weak-scaling leg")
        cfg5_line = cfg5_scaling_leg(env, rank, world, arith, args.steps, args.warmup, args.spinup_ms / 3.0,
                                     channels_override=2 * args.channels, n1_ms=args.n1_cfg5_ms)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_samples = float(world) * channels * bs * args.steps
        value = total_samples / elapsed / 1e6
        alg_bytes, rd_bytes = rx.algorithmic_bytes(bs)
        if q15:     # int16 slots: 4 B per complex sample in, 2 B per audio sample out; state bytes unchanged
            alg_bytes -= channels * (4 * bs + 2 * nout)
            rd_bytes -= channels * 4 * bs
        # roofline.achieved / frac come from ms_per_step -- the wall clock around the K timed steps (barrier + synchronize on both
        # sides), the figure the driver can check against its own clock; the HIP-event time of the same K steps (two events on the
        # kernels' stream, inside the library) is kept beside it, labelled
        k_ms = ev_ms if ev_ms is not None else ms_per_step
        achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9
        # flops the timed kernel executes: with one LO shared by all channels (the default shape) the NCO costs the
        # complex multiply only (6 flops per sample); the per-sample table-lerp sin/cos is the "per_channel" leg below
        shared_lo = rx.nco_path().startswith("shared") or "period 256" in rx.nco_path()
        fps = ch.flops_per_sample(spec, 6.0 if shared_lo else 20.0)
        fl = fps * channels * bs
        traffic = pmc_traffic(args.workload + ("_q15" if q15 else ""), args.arith, rx.kernel_name(), channels, bs,
                              dict(ch.WORKLOADS, **{k + "_q15": v for k, v in ch.WORKLOADS.items()}))
        split_kernel = arith in (sr.ARITH_SPLIT16, sr.ARITH_AUTO) and "split16" in rx.kernel_name()
        out = {
            "metric": "Msamples/s complex I/Q through full RX chain (whole job)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "spinup_ms": args.spinup_ms, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 in / out / accumulate; FIR multiplicands as exact f16 hi+lo pairs (~22 bits) on the matrix cores"
                      if split_kernel else "f32"),
            "data": "synthetic", "io": args.io,
            "config": {"workload": "%s: %d channels/GPU x %d complex samples/call, %s" % (
                           args.workload, channels, bs, WORKLOAD_TEXT[args.workload]),
                       "arith": {sr.ARITH_CMSIS: "cmsis-exact (mul,add)", sr.ARITH_FMA: "fma (<=1e-5 rel vs CMSIS)",
                                 sr.ARITH_SPLIT16: "split16 (f16 hi/lo x3 MFMA FIR, block floating point; <=1e-5 rel vs CMSIS on blocks within 12 dB of the input level, the others are counted: `guard`)",
                                 sr.ARITH_AUTO: "auto (split16 + bit-exact rerun of the channels under the parity guard: <=1e-5 rel vs CMSIS on every DSP block)"}[arith],
                       "kernel": rx.kernel_name(), "agc": "global" if args.global_gain else "per-channel",
                       "nco": rx.nco_path(),
                       "nco_note": ("headline shape of BASELINE cfg3: ONE NCO step for all channels, on the fs/256 grid -- the LO is computed once "
                                    "and lives in registers, the timed kernel evaluates no sin/cos; `other_nco_modes` times per-channel steps "
                                    "(arm_sin/cos_f32 per sample: what the cpu_baseline does) on the same data") if shared_lo else None,
                       "parallelism": "channels sharded x%d, no data-path collective" % world},
            "per_gpu_msamples_s": round(value / world, 2),
            "guard": dict(guard, ratio=0.25, note="DSP blocks of the timed steps under the parity guard (envelope < ratio x pass maximum); "
                                                  "auto: their channels were recomputed bit-exactly inside the call"),
            "guard_blocks": guard["blocks"],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "traffic_source": None if traffic is None else os.path.relpath(TRAFFIC_JSON, ROOT) + " (committed rocprofv3 --pmc passes of this kernel and shape; not measured in this run)",
                         "traffic_note": ("SELENITE_ARITH_AUTO: includes the handover-repair rows k_ssb_split16 writes in front of the decimator state "
                                          "(2 KB per channel and call for f32 slots, 1 KB for int16 slots -- DESIGN.md section 3); the algorithmic figure of SURVEY 8d "
                                          "does not count them") if (arith == sr.ARITH_AUTO and "k_ssb_split16" in rx.kernel_name()) else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "read_bytes_per_launch": rd_bytes,
                         "frac_source": "ms_per_step (wall clock around the timed steps; the HIP-event figure is frac_hip_events)",
                         "read_frac": round(rd_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "launch_ms_hip_events": round(k_ms, 4),
                         "frac_hip_events": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "launch_ms_median": None if each is None else round(float(each[len(each) // 2]), 4),
                         "launch_ms_min": None if each is None else round(float(each[0]), 4),
                         "launch_ms_p90": None if each is None else round(float(each[int(0.9 * len(each))]), 4),
                         "frac_median": None if each is None else round(alg_bytes / (float(each[len(each) // 2]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "fma_roof": {"achieved": round(fl / (k_ms * 1e-3) / 1e12, 2), "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(fl / (k_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4),
                         "flops_per_sample": fps,
                         "note": ("algorithmic f32 flops of the chain against the f32 vector / f32-MFMA peak (SURVEY 8d: cfg3 is "
                                  "FMA-bound in direct-form f32)" + ("; the split16 kernel executes the FIR on the f16 matrix pipe "
                                  "instead (three f16 MFMAs per product), so this roof does not bind it" if split_kernel else ""))},
        }
        if roof_ms is not None:
            r_med = float(roof_ms[len(roof_ms) // 2])
            out["streaming_roof"] = {
                "ms_per_launch": round(r_med, 4), "ms_min": round(float(roof_ms[0]), 4), "ms_p90": round(float(roof_ms[int(0.9 * len(roof_ms))]), 4),
                "launches": int(len(roof_ms)), "achieved": round(alg_bytes / (r_med * 1e-3) / 1e9, 1), "unit": "GB/s",
                "frac_of_peak": round(alg_bytes / (r_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "read_frac_of_peak": round(rd_bytes / (r_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": "k_stream_roof",
                "note": "the algorithmic bytes of one call (input in, audio out, per-channel state in and out) moved with the fused kernels' "
                        "access pattern -- persistent single-wave workgroups, 1 KB non-temporal wave loads, next pass prefetched -- and NO "
                        "arithmetic, timed in this run after the timed steps (median of per-launch HIP events): the floor of any kernel "
                        "with this traffic on this box at these clocks"}
            out["roofline"]["frac_of_streaming_roof"] = round(r_med / ms_per_step, 4)
            out["north_star_target"] = {
                "text": ">= 60 % of HBM-read roofline on the 256-tap FIR-dominated SSB chain at 1 GPU",
                "read_frac": out["roofline"]["read_frac"], "target": 0.60,
                "read_frac_of_the_no_arithmetic_kernel": out["streaming_roof"]["read_frac_of_peak"],
                "note": "read_frac counts READ bytes only (SURVEY.md 8d: 8 B per input sample + the state read) over ms_per_step against 8 TB/s; "
                        "the no-arithmetic streaming kernel of the same run sets what that figure can be at most for this traffic shape"}
        if world > 1 or env.dist is not None:
            out["dist"] = dist_block(env, devices, 1 if args.global_gain else 0, per_rank_ms, comm)
        if cfg5_line is not None:
            out["cfg5_weak_scaling"] = cfg5_line
        if args.n1_ms > 0:
            out["weak_scaling_efficiency"] = round(args.n1_ms / ms_per_step, 4)      # T_1 / T_N at fixed work per GPU, from the figure handed in
            out["n1_ms_per_step"] = args.n1_ms
        if world == 1 and not args.global_gain and not args.main_only:
            # the same workload in the other arithmetic contracts and with the general NCO flavours, outside the timed
            # region; every leg: own spin-up, >= 100 launches, median of per-launch HIP-event durations
            others = {}
            for nm, ar in (("split16", sr.ARITH_SPLIT16), ("auto", sr.ARITH_AUTO), ("fma", sr.ARITH_FMA), ("cmsis", sr.ARITH_CMSIS)):
                if ar == arith:
                    continue
                rx_x = make_rx(ar)
                others[nm] = leg(rx_x)
                rx_x.close()
            if arith == sr.ARITH_AUTO and "split16" in rx.kernel_name() and "hilb" not in rx.kernel_name():
                # what the handover repair of AUTO costs ("across a call boundary"): the same mode without the
                # mixed samples k_ssb_split16 keeps in front of the decimator state (2 KB per channel and call on this chain)
                rx_x = make_rx(sr.ARITH_AUTO)
                rx_x.set_handover_repair(False)
                others["auto_without_handover_repair"] = leg(rx_x)
                rx_x.close()
            others["note"] = ("auto: split16 + bit-exact rerun of guarded channels (handover repair on: the library default); split16: tolerance-based, guarded blocks only counted; "
                              "fma: bit-exact vs the oracle's fmaf restatement; cmsis: bit-exact (0 ULP) vs CMSIS-DSP 1.5.3 arithmetic")
            out["other_arith_modes"] = others
            if spec.nco and cfg_name == "cfg3" and args.nco == "default":
                frac = lambda r: round(alg_bytes / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                nm = {}
                raw = sr.ARITH_SPLIT16 if arith == sr.ARITH_AUTO else arith     # NCO cost alone: no reruns in these legs
                # every channel its own NCO step: arm_sin_f32 / arm_cos_f32 per sample inside the kernel (arm_sin_f32.c:72-119)
                rx_p = make_rx(raw, "per_channel")
                nm["per_channel"] = dict(leg(rx_p), note="every channel its own arbitrary NCO step; table-lerp sin/cos per sample in the kernel")
                rx_p.close()
                # every channel its own step ON the fs/256 grid (a channeliser): the channel's LO period is computed once per call
                rx_g = make_rx(raw, "per_channel_grid")
                nm["per_channel_grid"] = dict(leg(rx_g), note="every channel its own NCO step on the fs/256 grid (0 and fs/256 alternating, so the bench signal stays in band): LO of one 256-sample period computed per channel and call (arm_sin/cos_f32 arithmetic), held in registers")
                rx_g.close()
                if "registers" in rx.nco_path():
                    # the same shared LO read as a per-call table from L2 (what a step off the fs / 256 grid gets)
                    rx_t = make_rx(raw, "shared_table")
                    nm["shared_table"] = dict(leg(rx_t), note="one NCO step for all channels that is NOT a multiple of fs / 256: LO table computed per call, read from L2")
                    rx_t.close()
                for v in nm.values():
                    v["roofline_frac"] = frac(v)
                out["other_nco_modes"] = nm
                if arith == sr.ARITH_AUTO and not q15:
                    # what AUTO costs when the guard fires: per-channel steps anywhere on the fs/256 grid move most channels' tone out of
                    # the pass band (audio 20-45 dB under the input), the worst case for an output-relative bar
                    rx_a = make_rx(sr.ARITH_AUTO, "per_channel_grid_wide")
                    r = leg(rx_a)
                    rx_a.guard_clear()
                    rx_a.process_device(d_in.ptr, d_out.ptr, bs)
                    g1 = rx_a.guard_stats()
                    rx_a.close()
                    out["auto_stopband_cost"] = dict(r, roofline_frac=frac(r), rerun_fraction=round(g1["rerun_channel_calls"] / channels, 4),
                                                     note="SELENITE_ARITH_AUTO with per-channel steps anywhere on the fs/256 grid (--nco per_channel_grid_wide): the guarded fraction of the channels "
                                                          "is recomputed by the bit-exact kernel every call")
        if world == 1 and not args.global_gain and not args.main_only and args.workload == "cfg3" and not q15 and not args.block_size and args.arith == "auto" and args.nco == "default":
            # the other single-GPU BASELINE configurations, in the same driver-run line.  The headline's batch STAYS resident (2.7 GB
            # of 288): where hipMalloc puts a workload's buffers decides which of two modes the memory system runs cfg2 in -- kernel and plain copy
            # alike, profiles/r5/placement_probe.txt -- and behind a resident allocation it has been the faster one on two boxes of three (6 of 6 runs; the third: 0 of 2); each entry carries the
            # copy's time (streaming_roof_ms) so that the kernel's share of the figure can be told from the placement's
            wls = {}
            for nm, wq in (("cfg2", False), ("cfg4", False), ("cfg5", False), ("cfg3", True)):
                wls[nm + ("_q15" if wq else "")] = side_workload(nm, wq, arith, args.spinup_ms / 3.0, parity=not args.no_cpu_baseline, channels_override=args.channels)
            # cfg2 runs in one of two modes of the memory system depending on where its buffers land (kernel and plain copy alike): the entry above is
            # what two plain hipMalloc calls gave THIS process; the same workload with both buffers inside one allocation right behind it, so that the
            # line carries two placements and a reader can tell the placement's share from the kernel's (frac_of_streaming_roof stays put)
            one = side_workload("cfg2", False, arith, args.spinup_ms / 3.0, parity=False, channels_override=args.channels, one_allocation=True)
            pick = lambda w: {"ms_per_step": w["ms_per_step"], "value": w["value"], "frac": w["roofline"]["frac"],
                              "streaming_roof_ms": w["roofline"]["streaming_roof_ms"], "frac_of_streaming_roof": w["roofline"]["frac_of_streaming_roof"]}
            wls["cfg2"]["placement"] = {"separate_allocations": pick(wls["cfg2"]), "one_allocation": pick(one),
                                        "note": "separate_allocations = the entry itself (two hipMalloc calls); one_allocation = input and output inside one hipMalloc.  Which of "
                                                "the two modes a placement lands in differs from box to box and from process to process (profiles/r5/placement_probe.txt: "
                                                "one_allocation slow in 9 of 9 processes on one box; profiles/r6: both fast on another) -- the copy of the same bytes "
                                                "(streaming_roof_ms) moves with it; quote the slower of the two"}
            wls["note"] = ("every entry: its own instance and resident synthetic batch, own spin-up, median of 100 per-launch HIP-event durations; cfg2 = BASELINE's 4096 channels x "
                           "48 000 samples (one second per call); cfg4 = the CW chain (bit-exact in every arithmetic mode); cfg5 = the per-GPU shard of the weak-scaling "
                           "config (131 072 channels x 1024); cfg3_q15 = the headline with int16 slots in and out")
            out["workloads"] = wls
        if world == 1 and not args.no_cpu_baseline and not args.main_only:
            out["cpu_baseline"] = cpu_baseline(args.workload, ch.WORKLOADS)
            if not args.global_gain:
                out["parity"] = dict(parity_check(args.workload, ch.WORKLOADS, arith, q15), kernel=rx.kernel_name())
                out["parity_worst_rel"] = out["parity"]["worst_rel"]
        print(json.dumps(out), flush=True)

    rx.close()
    d_in.free()
    d_out.free()
    env.close()


if __name__ == "__main__":
    main()
