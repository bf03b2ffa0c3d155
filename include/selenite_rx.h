/*
 * selenite_rx.h -- C-ABI of the MI355X-native Selenite Lite RX DSP block path.
 *
 * Drop-in boundary.  Every entry point mirrors the CMSIS-DSP
 * "instance / init / process(S, pSrc, pDst, blockSize)" convention that the
 * reference vendors (Drivers/CMSIS/DSP/Include/arm_math.h:1175-1195 arm_fir_f32,
 * :1326-1344 arm_biquad_cascade_df1_f32, :3288-3312 arm_fir_decimate_f32) and sits
 * in the per-block callback slot of the firmware
 * (Core/Src/dsp_if.c:50-54,63-67 HAL_I2SEx_TxRx{Half,}CpltCallback, between the
 * int16 I/Q that DSP_In_Buff_Write (dsp_if.c:250-301) receives and the int16
 * audio DSP_Out_Buff_Read (dsp_if.c:204-219) hands back).
 *
 * Plain C: pointers and sizes only, no C++/torch types.  The library behind it
 * (libselenite_rx.so) is hand-written HIP for gfx950; there is NO CPU fallback:
 * selenite_rx_init() fails with SELENITE_RX_DEVICE_ERROR when no HIP device is
 * usable.
 *
 * Data layout (device and host views are the same):
 *   pSrcIQ    float  [channels][blockSize][2]   I,Q adjacent  (dsp_if.c:286-289 interleave)
 *   pDstAudio float  [channels][blockSize/decim]
 *   q15 variants: int16_t with the same shapes (arm_q15_to_float / arm_float_to_q15
 *   semantics, SupportFunctions/arm_q15_to_float.c:65-113, arm_float_to_q15.c:64-122).
 *
 * The chain that runs per channel and per DSP block is specified in DESIGN.md
 * ("Chain specification"); every step is one CMSIS-DSP primitive:
 *   NCO (arm_sin_f32/arm_cos_f32 + arm_cmplx_mult_cmplx_f32)
 *   -> arm_fir_decimate_f32 on the I and the Q rail
 *   -> arm_fir_f32 (delay taps) on I, arm_fir_f32 (Hilbert taps) on Q
 *   -> arm_sub_f32 / arm_add_f32 (USB / LSB)   [AM: arm_cmplx_mag_f32]
 *   -> arm_biquad_cascade_df1_f32 (CW narrow filter)
 *   -> arm_abs_f32 + arm_max_f32 -> gain law -> arm_scale_f32 (AGC)
 */
#ifndef SELENITE_RX_H
#define SELENITE_RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header (selenite_rx_abi_version() returns the library's).
 *   1  rounds 1-4: selenite_rx_config ends with agc_gain_init (112 bytes on LP64, the last 4 of them tail padding).
 *   2  selenite_rx_config grows to 120 bytes: q15_rounding (in the former padding), abi_version, reserved -- same for selenite_tx_config
 *      (96 -> 104 bytes).  selenite_rx_init / selenite_tx_init tell the two layouts apart by struct_size, CMSIS-style (the caller owns
 *      the struct, the library reads only what that caller's header declared, cf. arm_math.h:3272-3312): a version-1 caller
 *      (struct_size = SELENITE_RX_CONFIG_SIZE_V1) keeps working unchanged -- the bytes behind agc_gain_init are NOT read, whatever
 *      they hold, and int16 output truncates as it always did; a version-2 caller (struct_size = sizeof) must set abi_version = 2. */
#define SELENITE_RX_ABI_VERSION 2
#define SELENITE_RX_CONFIG_SIZE_V1 112u

/* Status codes: numerically identical to arm_status (arm_math.h:399-408). */
#define SELENITE_RX_SUCCESS          0   /* ARM_MATH_SUCCESS          */
#define SELENITE_RX_ARGUMENT_ERROR (-1)  /* ARM_MATH_ARGUMENT_ERROR   */
#define SELENITE_RX_LENGTH_ERROR   (-2)  /* ARM_MATH_LENGTH_ERROR     */
#define SELENITE_RX_NANINF         (-4)  /* ARM_MATH_NANINF (arm_math.h:405): a process call produced NaN / Inf audio -- latched by
                                           * selenite_rx_sync() and the host-pointer process calls (every fused kernel) */
#define SELENITE_RX_DEVICE_ERROR   (-7)  /* outside arm_status: HIP device/runtime failure */

/* Demodulator modes: the values of the firmware's Mode enum (Core/Inc/rxtx_if.h:33-43),
 * i.e. the byte DSP_Set_Mode() (Core/Src/dsp_if.c:367-370) receives from the CAT parser. */
#define SELENITE_MODE_LSB 0x00
#define SELENITE_MODE_USB 0x01
#define SELENITE_MODE_CW  0x02
#define SELENITE_MODE_CWR 0x03
#define SELENITE_MODE_AM  0x04
#define SELENITE_MODE_FM  0x08   /* Build-defined (the reference has no FM demodulator, CMSIS-DSP 1.5.3 no arctangent):
                                  * audio[n] = angle(z[n] * conj(z[n-1])) / pi on the decimated I/Q -- arm_cmplx_conj_f32,
                                  * arm_cmplx_mult_cmplx_f32, a stated arctangent, arm_scale_f32.  The
                                  * sample in front of a block comes from the delay lines of the FIR pair, which this mode keeps
                                  * running without evaluating the taps: needs nh_taps >= 2 (ARGUMENT_ERROR otherwise).
                                  * SELENITE_ARITH_SPLIT16 runs FM as _FMA; _AUTO guards it on min|z| x max|audio|. */
#define SELENITE_MODE_DIG 0x0A   /* = USB */
#define SELENITE_MODE_PKT 0x0C   /* = LSB */

/* Arithmetic contract of the MAC loops (DESIGN.md "Arithmetic modes"). */
#define SELENITE_ARITH_CMSIS 0   /* product rounded, then sum rounded: bit-exact vs CMSIS-DSP 1.5.3 C code */
#define SELENITE_ARITH_FMA   1   /* same operation order, the multiply-add of the FIR tap loops fused (fmaf);
                                    NCO, biquad recurrence and AGC keep the reference rounding.  Bit-exact
                                    vs the oracle's fmaf restatement, <=1e-5 relative vs CMSIS */
#define SELENITE_ARITH_SPLIT16 2 /* the many-tap FIR of the shape as a split-precision matrix product on the 16-bit matrix
                                    cores (decimator of the /2, /4 and /8 shapes; Hilbert FIR of the no-decimator shapes; TX
                                    interpolator): samples and taps split into f16 hi + lo parts, the three significant
                                    products (hi*hi, hi*lo, lo*hi) accumulated in f32 by MFMA, with a block exponent taken
                                    from the data of every pass (any amplitude a float can hold).  NOT bit-exact against any
                                    CPU order.  Guarantee: per DSP block max|out - ref| <= 1e-5 max|ref| against the CMSIS
                                    chain wherever the block's audio is within 12 dB of the largest sample its matrix product
                                    saw; every DSP block outside that zone is COUNTED (selenite_rx_guard_stats), and there the
                                    figure is the split product's error against the INPUT level (~1e-6 of it).  Everything
                                    else in the chain, and every configuration without such a kernel, runs as _FMA.
                                    Streaming filter state stays exact f32. */
#define SELENITE_ARITH_AUTO 3    /* _SPLIT16 with the conditional zone closed -- the default of the benchmarks.  A channel with
                                    a guarded DSP block in a call (selenite_rx_set_guard_ratio) is recomputed for that call,
                                    from its pre-call streaming state, by the bit-exact _CMSIS kernel, on the device, inside
                                    the same process call.  Guarantee: per DSP block max|out - ref| <= 1e-5 max|ref| against
                                    the CMSIS chain on EVERY block of every call, for any input, call length and mode switch
                                    (unguarded blocks by the split product's accuracy, recomputed channels with 0 ULP;
                                    selenite_rx_set_handover_repair keeps the recomputation exact across call boundaries).
                                    int16 slots: the bar holds for the float audio in front of arm_float_to_q15; the int16
                                    words are within 1 LSB + 1e-5 x (block maximum of that float audio) x 32768 of the
                                    reference's (a float inside the bar can still flip the truncation; only _CMSIS / _FMA
                                    are bit-exact there).
                                    Split-precision kernels exist for every decimator of 2 .. 256 taps (even count) by 2 or
                                    by 4 in front of a 31- / 63- / 127-tap type-III pair, by 8 in front of a 63-tap one, and
                                    for those pairs alone; other configurations run as _CMSIS. */

typedef struct selenite_rx_config {
    uint32_t struct_size;     /* = sizeof(selenite_rx_config) (a caller built against the version-1 header passes SELENITE_RX_CONFIG_SIZE_V1) */
    uint32_t channels;        /* C: independent I/Q channels handled by this instance */
    uint32_t block;           /* DSP block: complex input samples per CMSIS call / AGC update */
    uint32_t decim;           /* M (arm_fir_decimate_instance_f32.M); 1 = no decimator */
    uint32_t nd_taps;         /* decimator taps (numTaps); 0 = decimator bypassed (requires decim==1) */
    uint32_t nh_taps;         /* taps of the Hilbert / delay FIR pair; 0 = no FIR pair */
    uint32_t n_biquad;        /* DF1 biquad stages applied in CW/CWR; 0 = none */
    uint32_t arith;           /* SELENITE_ARITH_* */
    uint8_t  mode;            /* SELENITE_MODE_* */
    uint8_t  nco_enable;      /* 1 = quadrature NCO mix in front of the chain */
    uint8_t  agc_enable;      /* 1 = AGC at the end of the chain */
    uint8_t  agc_global;      /* 1 = one gain from the envelope maximum over ALL channels (and ranks) */
    uint32_t nco_step_all;    /* phase step per input sample (2^32 = one turn) when nco_step == NULL */
    const float    *dec_coeffs;    /* [nd_taps]  CMSIS order {b[N-1] .. b[0]} (arm_fir_decimate_f32.c:64-69) */
    const float    *hilb_coeffs;   /* [nh_taps]  arm_fir_f32 taps applied to the Q rail */
    const float    *delay_coeffs;  /* [nh_taps]  arm_fir_f32 taps applied to the I rail */
    const float    *biquad_coeffs; /* [5*n_biquad] {b0,b1,b2,a1,a2} per stage, feedback ADDED
                                      (arm_biquad_cascade_df1_f32.c:52-63) */
    const uint32_t *nco_step;      /* [channels] per-channel phase step, or NULL */
    float agc_target;         /* wanted peak level of the audio block */
    float agc_attack;         /* one-pole rate when the gain has to fall */
    float agc_decay;          /* one-pole rate when the gain may rise */
    float agc_gain_min;
    float agc_gain_max;
    float agc_env_floor;      /* envelope is clamped from below to this before target/env */
    float agc_gain_init;      /* gain before the first block */
    uint32_t q15_rounding;    /* int16 output (arm_float_to_q15): 0 = truncate -- the reference's code as the firmware builds it
                                 (ARM_MATH_ROUNDING undefined, arm_float_to_q15.c:117) --, 1 = the ARM_MATH_ROUNDING variant
                                 (arm_float_to_q15.c:90-101: +-0.5 in float before the truncation); any other value is an ARGUMENT_ERROR.
                                 ABI version 2: read only when struct_size == sizeof(selenite_rx_config) (it sits where version 1 had padding). */
    uint32_t abi_version;     /* ABI version 2: = SELENITE_RX_ABI_VERSION (anything else with the version-2 struct_size: ARGUMENT_ERROR) */
    uint32_t reserved;        /* 0 */
} selenite_rx_config;

/* Host-side view of the per-channel streaming state (what CMSIS keeps in pState).
 * Any pointer may be NULL (skipped).  Layouts:
 *   dec_state  [channels][2][nd_taps-1]   rail 0 = I, rail 1 = Q; oldest sample first
 *                                         (arm_fir_decimate_f32.c:396-426 copy-back order)
 *   fir_state  [channels][2][nh_taps-1]   rail 0 = delay FIR (I), rail 1 = Hilbert FIR (Q)
 *                                         (arm_fir_f32.c:947-978)
 *   biq_state  [channels][n_biquad][4]    {x[n-1], x[n-2], y[n-1], y[n-2]}
 *                                         (arm_biquad_cascade_df1_f32.c:319-322)
 *   agc_gain   [channels]
 *   nco_phase  [channels]                 uint32 phase accumulator
 */
typedef struct selenite_rx_state_view {
    float    *dec_state;
    float    *fir_state;
    float    *biq_state;
    float    *agc_gain;
    uint32_t *nco_phase;
} selenite_rx_state_view;

typedef struct selenite_rx_instance selenite_rx_instance;  /* opaque; state lives in HBM */

/* ---- instance life cycle ------------------------------------------------------------- */

/* Mirrors arm_fir_decimate_init_f32 (FilteringFunctions/arm_fir_decimate_init_f32.c:63-101):
 * validates (block % decim != 0 -> SELENITE_RX_LENGTH_ERROR), clears all state.  Coefficient
 * arrays are copied to the device; the caller may free them afterwards.  Uses the calling
 * thread's current HIP device.  On failure *S is set to NULL. */
int  selenite_rx_init(selenite_rx_instance **S, const selenite_rx_config *cfg);
void selenite_rx_free(selenite_rx_instance *S);

/* Mirrors DSP_Set_Mode(uint8_t) (Core/Src/dsp_if.c:367-370).  Filter state is kept. */
int  selenite_rx_set_mode(selenite_rx_instance *S, uint8_t mode);

/* Sticky status of the instance: first error raised by a process call (they return void, as
 * their CMSIS counterparts do), or SELENITE_RX_SUCCESS. */
int  selenite_rx_status(const selenite_rx_instance *S);
/* Human-readable text for the last error of this thread / instance (never NULL). */
const char *selenite_rx_error_string(const selenite_rx_instance *S);

/* ---- the per-block process call ------------------------------------------------------- */

/* Host buffers, synchronous: the literal "float* I/Q in, float* audio out, blockSize" call of the slot.
 * blockSize = complex input samples per channel in this call; it must be a non-zero multiple of cfg.block
 * (several DSP blocks are processed back to back exactly as successive CMSIS calls would).  The call is cut into
 * channel chunks and pipelined over PCIe (H2D of chunk k+1 || kernels of chunk k || D2H of chunk k-1, two
 * device buffers each way, no allocation per call).  Page-locked caller memory (selenite_rx_host_alloc,
 * selenite_rx_host_register) is the DMA source / target itself; pageable memory is staged through the
 * library's pinned buffers by a few host threads.  Same bits as the _device call on the same data. */
void selenite_rx_process_f32(selenite_rx_instance *S, const float *pSrcIQ,
                             float *pDstAudio, uint32_t blockSize);

/* Page-locked host memory for the buffers of the host-pointer calls: allocate it, or register memory the caller
 * already owns (the firmware-side analogue is a DMA-capable buffer).  Optional: unregistered memory works, slower. */
void *selenite_rx_host_alloc(size_t bytes);
void  selenite_rx_host_free(void *hptr);
int   selenite_rx_host_register(void *hptr, size_t bytes);
int   selenite_rx_host_unregister(void *hptr);

/* Device buffers (HBM-resident), asynchronous on the instance's stream. */
void selenite_rx_process_f32_device(selenite_rx_instance *S, const float *dSrcIQ,
                                    float *dDstAudio, uint32_t blockSize);

/* int16 wire format of the slot (dsp_if.c:286-289): q15 in, q15 out; conversion is fused into
 * the kernels' load and store (arm_q15_to_float: /32768.0f; arm_float_to_q15: *32768.0f,
 * truncate, saturate). */
void selenite_rx_process_q15(selenite_rx_instance *S, const int16_t *pSrcIQ,
                             int16_t *pDstAudio, uint32_t blockSize);
void selenite_rx_process_q15_device(selenite_rx_instance *S, const int16_t *dSrcIQ,
                                    int16_t *dDstAudio, uint32_t blockSize);

/* Global-gain AGC (cfg.agc_global) split at the only cross-GPU exchange point:
 *   phase 1 runs the chain up to the un-scaled audio and leaves, per DSP block of this call,
 *           max|audio| over this instance's channels in dEnv[blockSize/cfg.block] (device);
 *   the caller all-reduces dEnv with MAX across ranks (RCCL) -- or not, single GPU;
 *   phase 2 runs the gain law on dEnv and scales dDstAudio in place.
 * selenite_rx_process_f32_device() on an agc_global instance = phase 1 + phase 2. */
void selenite_rx_global_phase1_device(selenite_rx_instance *S, const float *dSrcIQ,
                                      float *dDstAudio, float *dEnv, uint32_t blockSize);
void selenite_rx_global_phase2_device(selenite_rx_instance *S, float *dDstAudio,
                                      const float *dEnv, uint32_t blockSize);

/* The same in one call for a plain C host: phase 1, ncclAllReduce(ncclMax) of the blockSize / cfg.block envelopes
 * over `rccl_comm` (an ncclComm_t whose rank runs on this instance's device; NULL = single rank), phase 2, all
 * enqueued on the instance's stream.  RCCL is resolved at run time (no link dependency).  Returns the status. */
int selenite_rx_global_process_f32_device(selenite_rx_instance *S, const float *dSrcIQ, float *dDstAudio,
                                          uint32_t blockSize, void *rccl_comm);

/* ---- parity guard of the split-precision arithmetic (SELENITE_ARITH_SPLIT16 / _AUTO) -------------------------- */

/* A DSP block is GUARDED when max |audio| of the block (before the AGC) is below `ratio` x the largest |component| of the mixed
 * samples whose split-precision error can reach it: those its pass of the matrix product held (new samples and FIR history); for
 * the blocks that still read FIR-pair history of the pass before (the first nh_taps - 1 audio samples of a pass) also that pass's;
 * at a call's start also the level the call before left.  FM: guarded when min|z| x max|audio| is below `ratio` x that maximum (the
 * discriminator's error is |dz| / (pi |z|)).  Default ratio 0.25 (-12 dB); 0 disables the guard, +inf guards every block with
 * non-zero input (SELENITE_ARITH_AUTO then recomputes every channel bit-exactly: a test hook).  Takes effect with the next call.
 * SELENITE_ARITH_AUTO recomputes a channel that owns a guarded block with the bit-exact kernel, inside the same call, and then HOLDS
 * it there: the matrix kernel skips the channel in the following calls (a channel whose pass band is empty call after call costs
 * the bit-exact kernel's time, not both kernels') until two calls in a row show no block under 1.25 x ratio. */
int selenite_rx_set_guard_ratio(selenite_rx_instance *S, float ratio);
/* Counters since init / the last selenite_rx_guard_clear (any pointer may be NULL); drains the instance's stream:
 *   guard_blocks         DSP blocks guarded
 *   guard_channel_calls  (channel, process call) pairs with at least one guarded block
 *                        (SELENITE_ARITH_AUTO: plus every call of a channel the bit-exact kernel holds)
 *   rerun_channel_calls  (channel, call) pairs SELENITE_ARITH_AUTO computed with the bit-exact kernel (0 for _SPLIT16) */
int selenite_rx_guard_stats(selenite_rx_instance *S, uint64_t *guard_blocks, uint64_t *guard_channel_calls,
                            uint64_t *rerun_channel_calls);
/* per_channel[channels]: guarded DSP blocks of every channel since init / the last clear (sticky per-channel view). */
int selenite_rx_guard_channels(selenite_rx_instance *S, uint32_t *per_channel);
/* Diagnostic: per_channel[channels] = the word SELENITE_ARITH_AUTO keeps per channel (bit 0: recompute pending; bits 1-2: what the
 * call before left the streaming state as -- 0 exact, 1 matrix kernel with the samples for the repair, 2 without; bit 5: the channel is
 * HELD by the bit-exact kernel, bit 6: its last call there was clean; bits 8-31: level of the last pass).  Zeros in the other modes. */
int selenite_rx_auto_words(selenite_rx_instance *S, uint32_t *per_channel);
/* SELENITE_ARITH_AUTO across calls.  A channel the previous call left on the matrix kernel carries a FIR-pair history (the last
 * nh_taps - 1 decimated samples) of split16 precision; if THIS call has to be recomputed for it, its first blocks would start from
 * that history.  Handover repair (default ON): the matrix kernel also leaves the exact mixed samples in front of the decimator state
 * behind (decim * (nh_taps - 1) samples per channel and call, rounded up to whole quads of audio samples: 2 KB for the cfg3 chain;
 * two buffers -- 4 KB of device memory per channel, allocated by the first call that needs them and released when the repair is
 * switched off), and the recomputation derives the FIR-pair history from them in exact arithmetic first: a recomputed call is CMSIS
 * bit for bit from its first sample (apart from the gain the previous call's AGC left: ~1e-6 relative).  A call too short to hold
 * those samples (under nd_taps + decim * (nh_taps - 1) per channel: the firmware's literal one-slot callback) runs on the bit-exact
 * kernel in _AUTO, and the history is repaired in front of an AM call and in front of the CW / generic kernels -- so with the repair
 * ON no block ever starts from a history of split16 precision: selenite_rx_guard_handover stays 0 and the _AUTO guarantee is
 * unconditional.  Cost: those bytes (2 % of the headline at 4096 samples per call).
 * OFF (a diagnostic: what the repair is worth): nothing is kept; guarded blocks inside the reach of the history behind a call that
 * stayed on the matrix kernel carry the error of a guarded block of raw _SPLIT16 and are COUNTED (selenite_rx_guard_handover): */
int selenite_rx_set_handover_repair(selenite_rx_instance *S, int on);
/* SELENITE_ARITH_AUTO, where the recomputation runs.  launches = 1 (default): on the no-decimator shapes (k_hilb_split16: one channel per
 * workgroup) the workgroup that guarded a channel recomputes it itself, with the body of the bit-exact kernel -- one launch per call.
 * launches = 3: always in two launches behind the matrix kernel (a dense list of the channels; what the decimating shapes do anyway).
 * The RESULT does not depend on the choice: which arithmetic serves a channel is decided by the channel's word alone; both forms run
 * the same code on it and give the same bits (tests/test_gpu_auto_forms.py). */
int selenite_rx_set_auto_launches(selenite_rx_instance *S, int launches);
/* Diagnostic: the form the last SELENITE_ARITH_AUTO call on a matrix kernel took, 1 or 3 (0: none yet). */
int selenite_rx_auto_launches_last(const selenite_rx_instance *S);
int selenite_rx_guard_handover(selenite_rx_instance *S, uint64_t *handover_blocks);
int selenite_rx_guard_clear(selenite_rx_instance *S);

/* ---- streams, state, memory ----------------------------------------------------------- */

/* hipStream_t passed as void*; NULL = the library-owned stream created at init. */
int  selenite_rx_set_stream(selenite_rx_instance *S, void *hip_stream);
int  selenite_rx_sync(selenite_rx_instance *S);

int  selenite_rx_get_state(selenite_rx_instance *S, const selenite_rx_state_view *dst);
int  selenite_rx_set_state(selenite_rx_instance *S, const selenite_rx_state_view *src);
int  selenite_rx_reset(selenite_rx_instance *S);      /* state back to the post-init values */

void *selenite_rx_device_alloc(size_t bytes);          /* hipMalloc; NULL on failure */
void  selenite_rx_device_free(void *dptr);
int   selenite_rx_memcpy_h2d(void *dptr, const void *hptr, size_t bytes);
int   selenite_rx_memcpy_d2h(void *hptr, const void *dptr, size_t bytes);
int   selenite_rx_device_count(void);                  /* number of HIP devices, 0 if none */
int   selenite_rx_set_device(int ordinal);             /* hipSetDevice for the calling thread */

/* ---- measurement support (bench.py) -------------------------------------- */

/* Synthetic I/Q of SURVEY.md 8d: per channel 3 complex tones + uniform noise, integer phase
 * accumulators and table-lerp sin/cos only, so host and device produce identical bits.
 * Fills iq[nch][nsamp][2] for channels first_channel .. first_channel+nch-1 and samples
 * first_sample .. first_sample+nsamp-1. */
void selenite_rx_synth_iq_host(float *iq, uint32_t first_channel, uint32_t nch,
                               uint64_t first_sample, uint32_t nsamp, uint64_t seed);
int  selenite_rx_synth_iq_device(selenite_rx_instance *S, float *dIQ, uint32_t first_channel,
                                 uint32_t nch, uint64_t first_sample, uint32_t nsamp,
                                 uint64_t seed);

/* Runs `iters` back-to-back process_f32_device calls bracketed by HIP events recorded on the
 * instance's stream; returns the mean milliseconds per call in *ms_per_call (state advances
 * as in normal streaming).  This is the live per-launch duration bench.py reports. */
int  selenite_rx_time_process_device(selenite_rx_instance *S, const float *dSrcIQ,
                                     float *dDstAudio, uint32_t blockSize, uint32_t iters,
                                     float *ms_per_call);
/* The same over the int16 slot format (selenite_rx_process_q15_device calls). */
int  selenite_rx_time_process_q15_device(selenite_rx_instance *S, const int16_t *dSrcIQ,
                                         int16_t *dDstAudio, uint32_t blockSize, uint32_t iters,
                                         float *ms_per_call);

/* The same with a HIP event between consecutive calls: ms_each[iters] receives the duration of every call (SURVEY.md 8d asks
 * for the median of >= 20 launches).  q15 != 0: the buffers are the int16 slot format.  The extra events cost a little
 * overlap between consecutive launches; bench.py reports both this median and the plain mean above. */
int  selenite_rx_time_process_each_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                          uint32_t iters, float *ms_each, int q15);
/* The streaming roof of THIS instance's call, measured: `iters` launches of a kernel that moves exactly the algorithmic bytes of one
 * process call of blockSize samples (input in, audio out, per-channel state in and out -- a scratch copy, the instance's
 * state is not touched) with the access pattern of the fused kernels (persistent single-wave workgroups, 1 KB non-temporal wave loads,
 * next pass prefetched) and NO arithmetic; ms_each[iters] receives the per-launch durations.  dDstAudio is overwritten with junk.
 * bench.py reports the DSP kernels as a fraction of this floor beside the fraction of the nominal 8 TB/s. */
int  selenite_rx_time_streaming_roof_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                            uint32_t iters, float *ms_each, int q15);
/* The same for the shapes of the systolic CW kernel (k_cw_fused: CW / CWR, no FIR stages, 2 / 4 / 8 biquad sections, DSP blocks of 128 / 256 /
 * 512), whose fetch pattern is its own: one wavefront streams 64 / n_biquad channel rows at once, 1 KB of a row per load.  The kernel timed here
 * issues exactly those bursts and stores (same descriptors, same number of bursts in flight, same residency) and `work` dependent vector
 * instructions per chunk where the biquad steps are -- 0: what the pattern alone costs.  Other configurations: SELENITE_RX_ARGUMENT_ERROR. */
int  selenite_rx_time_pattern_roof_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                          uint32_t iters, float *ms_each, int q15, uint32_t work);

/* ---- kernel-selection overrides (tests, diagnosis) -------------------------------------
 * Process-wide words, NOT configuration: results never depend on them.  The test suite uses them to reach every product path (the generic
 * per-stage kernels, the NCO flavours, both grids of SELENITE_ARITH_AUTO's rerun pass).  Read by selenite_rx_init / selenite_tx_init when an
 * instance is planned (SELENITE_RX_OPT_RERUN_GRID: by every rerun launch).  The library reads no environment variable for any of this; its
 * only environment setting is SELENITE_RX_HOST_CHUNK_MB (chunk of the host-pointer pipeline, default 32). */
#define SELENITE_RX_OPT_FORCE_GENERIC     0   /* 1: RX instances created from now on use the generic per-stage kernels only */
#define SELENITE_RX_OPT_NO_SHARED_LO      1   /* 1: per-channel NCO in the kernel even when all channels share step and phase */
#define SELENITE_RX_OPT_NO_PERIODIC_LO    2   /* 1: a shared LO always as a per-call table, never a 256-sample period in registers (RX and TX) */
#define SELENITE_RX_OPT_RERUN_GRID        3   /* workgroups of SELENITE_ARITH_AUTO's rerun pass (1 .. 2^20); 0: sized from the last call's list */
#define SELENITE_RX_OPT_TX_FORCE_GENERIC  4   /* 1: TX instances created from now on use the generic kernels only */
#define SELENITE_RX_OPT_CW_GRID           5   /* workgroups of the systolic CW kernel (each takes the channel groups b, b + grid, ...; 1 .. 2^20);
                                               * 0: one per group, or a whole share of groups each where they divide evenly over the device */
#define SELENITE_RX_OPT_COUNT             6
int      selenite_rx_set_plan_option(int option, uint32_t value);   /* SELENITE_RX_ARGUMENT_ERROR: unknown option / value out of range */
uint32_t selenite_rx_get_plan_option(int option);
/* PCI bus id ("0000:05:00.0") of HIP device `ordinal` into buf; bench.py lists the devices of the ranks with it. */
int  selenite_rx_device_pci_bus_id(int ordinal, char *buf, size_t len);

/* Name of the kernel variant process_f32_device dispatches to for this instance
 * (e.g. "rx_ssb_fused<256,4,63>" or "generic"); for logs and profiles. */
const char *selenite_rx_kernel_name(const selenite_rx_instance *S);
/* How the NCO of the next call is served: "off", "per-channel arm_sin/cos_f32 in the kernel" (steps or phases differ
 * between channels: arm_sin_f32.c:72-119 per sample), "shared LO table per call" (one step, one phase: the table is
 * computed once per call with the same arm_sin/cos arithmetic), or "shared LO, period 256 samples, held in registers"
 * (same table; a step that is a multiple of 2^24 repeats it every 256 samples).  All three give identical results. */
const char *selenite_rx_nco_path(const selenite_rx_instance *S);

/* Algorithmic HBM bytes of one process call (SURVEY.md 8d formula):
 *   channels * (8*blockSize + 4*blockSize/decim + S_in + S_out).
 * read_bytes (optional) receives the read-only part 8*blockSize + S_in per channel. */
uint64_t selenite_rx_algorithmic_bytes(const selenite_rx_config *cfg, uint32_t blockSize,
                                       uint64_t *read_bytes);

/* ---- coefficient design helpers (host only, double precision -> float) ------------------ */
/* All write CMSIS coefficient order.  They are conveniences for callers and tests; the chain
 * itself only ever sees the arrays in selenite_rx_config. */

/* Hamming-windowed sinc low-pass, unity DC gain; cutoff as a fraction of the INPUT sample rate. */
int selenite_rx_design_lowpass(float *coeffs, uint32_t num_taps, double cutoff);
/* Hamming-windowed type-III Hilbert transformer (odd num_taps) and its matched delay
 * (unit impulse at (num_taps-1)/2). */
int selenite_rx_design_hilbert(float *hilb, float *delay, uint32_t num_taps);
/* n_stages identical RBJ constant-peak band-pass sections centred on f0 (fraction of the
 * sample rate) with quality factor q; coefficients in CMSIS sign convention (+a1, +a2). */
int selenite_rx_design_bandpass(float *coeffs, uint32_t n_stages, double f0, double q);

int selenite_rx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SELENITE_RX_H */
