// rx_internal.h -- instance layout and kernel-launch interfaces shared by the translation units
// of libselenite_rx.so.  Not part of the C-ABI (include/selenite_rx.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/selenite_rx.h"
#include "rx_device.h"
#include "rx_diag.h"

namespace srx {

// Everything a kernel needs, passed by value (kernarg segment -> SGPRs).
struct RxParams {
    uint32_t channels;
    uint32_t block;        // DSP block (input samples)
    uint32_t decim;        // M
    uint32_t nd;           // decimator taps (0 = bypass)
    uint32_t nh;           // Hilbert / delay taps (0 = none)
    uint32_t nbiq;         // biquad stages
    uint32_t mode;         // SELENITE_MODE_*
    uint32_t nco;          // 0 = off, 1 = per-channel LO computed in the kernel, 2 = shared LO table `lo`
    uint32_t agc;          // AGC enabled
    uint32_t block_size;   // input samples per channel in this call
    uint32_t nout;         // block_size / decim
    uint32_t in_stride;    // complex samples between consecutive channels in the source buffer (>= block_size)
    uint32_t out_stride;   // audio samples between consecutive channels in the destination buffer (>= nout)
    uint32_t pass_out;     // generic front kernel: decimated outputs per pass
    uint32_t q15_round;    // int16 output: 1 = the ARM_MATH_ROUNDING variant of arm_float_to_q15 (arm_float_to_q15.c:90-101), 0 = truncation (the firmware's build)
    uint32_t out_cached;   // 1: the audio this launch writes is read back by a later kernel of the same call (global gain, phase 1):
                           // default store policy; 0: written once -- non-temporal stores
    const float *dec_c, *hilb_c, *delay_c, *biq_c, *sintab;
    const float2 *lo;      // nco == 2: LO[n] = (cos, -sin) for the samples of this call (all channels share it)
    uint32_t lo_period;    // nco == 2: 256 when LO[n + 256] == LO[n] for every n (NCO step a multiple of 2^24), else 0;
                           // nco == 1: 256 when EVERY channel's step is a multiple of 2^24 (each channel's own LO has that period)
    const uint32_t *step;
    uint32_t *phase;
    float *dec_state;      // [C][2][nd-1]
    float *fir_state;      // [C][2][nh-1]
    float *biq_state;      // [C][nbiq][4]
    float *gain;           // [C]
    float *env_part;       // global gain, phase 1 (k_ssb_split16, 16-lane DSP blocks, whole passes): max |audio| of every DSP block,
                           // [channels][block_size / block]; NULL = not wanted
    uint32_t *flags;       // device flag / counter words (kFlag* below): [0] bit 0 = a kernel produced non-finite audio (ARM_MATH_NANINF)
    // ---- parity guard of the split-precision kernels ----
    // A DSP block is GUARDED when max |audio| (before the AGC) < guard_ratio * the largest |component| of the samples its
    // pass's matrix product saw: the region where the split product and CMSIS, two f32-class results with independent
    // rounding noise of ~1e-7 of the INPUT, may differ by more than 1e-5 of the (small) block maximum.
    float guard_ratio;
    uint32_t *guard_ch;    // [channels] sticky count of guarded DSP blocks per channel, or NULL
    uint32_t *guard_calls; // [channels] sticky count of process calls in which the channel had a guarded block, or NULL
    uint32_t *guard_hand;  // [channels] sticky count of "handover" blocks (SELENITE_ARITH_AUTO, k_ssb_split16): guarded blocks inside the
                           // Hilbert-pair history of a call whose predecessor left the channel's state in split16 precision, or NULL
    // SELENITE_ARITH_AUTO: a channel with a guarded block in this call keeps its pre-call streaming state (the split16
    // kernel does not write it back) and raises rerun_flag[channel] (every channel's flag is rewritten every call: plain
    // stores, no atomics -- 48 k atomics on one counter cost 0.7 ms per launch when most channels were guarded); a second
    // launch of the bit-exact kernel (chan_flags = those flags) recomputes the call for the flagged channels -- audio and
    // state -- in the CMSIS arithmetic
    uint32_t *rerun_flag;        // [channels] or NULL (plain SPLIT16: count only).  Word of a channel (SELENITE_ARITH_AUTO):
                                 //   bit 0      recompute this channel's current call in exact arithmetic (kFlagRerun)
                                 //   bits 1-2   provenance of the channel's streaming state (kProv*): what the call before left
                                 //   bit 3      which of the two hist_ext buffers holds the samples in front of that state
                                 //   bit 4      that row holds raw int16 samples (kExtQ15)
                                 //   bit 5      hysteresis (kFlagHold): the exact kernel serves this channel DIRECTLY -- the matrix kernel skips
                                 //              it -- until two calls in a row show no block near the guard ratio (bit 6: one has) (round 4)
                                 //   bits 8-31  the largest |component| the LAST pass of the call before held in its images (the upper 24 bits
                                 //              of the float's bit pattern, rounded up): the first blocks of this call still see Hilbert-pair
                                 //              history computed from those samples, so their guard threshold covers them too (kLvlMask)
    // k_ssb_split16, SELENITE_ARITH_AUTO: the mixed samples in front of the decimator state, hist_ext[buf][channel][ext_len] (I, Q),
    // positions [E - (nd - 1) - ext_len, E - (nd - 1)) of the stream that ends at E -- what it takes to recompute the Hilbert-pair
    // history (the last HH4 decimated samples) in exact arithmetic when the NEXT call has to be rerun (k_hist_exact)
    float2 *hist_ext;            // or NULL
    uint32_t ext_len;            // decim * HH4
    size_t ext_buf_stride;       // elements between the two buffers
    uint32_t *chan_flags;  // exact kernels as the rerun pass of SELENITE_ARITH_AUTO: the channel words (= rerun_flag of the matrix kernel's launch)
    // ... and the channels to recompute as a DENSE list (round 4): k_hist_exact, in front of the rerun pass, appends every channel whose
    // rerun bit is up (one atomic per 16-channel window that has any) -- workgroup b of the rerun pass then takes entries b, b + grid, ...:
    // an even share whatever the pattern of flagged channels (walking the words in windows gave the slowest workgroup ~32 channels when
    // 80 % were flagged: as long as recomputing everything).  Two counters alternate between calls: a call's prepare kernel zeroes
    // the one the NEXT call will count in.
    uint32_t *chan_list;         // [channels]
    uint32_t *chan_count;        // entries in chan_list (device)
    uint32_t *chan_count_next;   // zeroed by this call's prepare kernel
    uint32_t *rerun_par_host;    // HOST word (the instance's): which of the two counters the next prepare kernel counts in -- read and
                                 // flipped where that kernel is launched (rx_fused.hip: launch_shape), so a call that never launches it
                                 // (a short call on the bit-exact kernel) leaves the pair in step
    uint32_t auto_inline;        // SELENITE_ARITH_AUTO: 1 = ONE launch where the matrix kernel can recompute the channel it flagged itself (k_hilb_split16:
                                 // FusedArgs::inl); 0 = always k_hist_exact + the rerun pass behind it.  The bits of the result do not depend on it.
    uint32_t *form_host;         // HOST word (the instance's): the form the last SELENITE_ARITH_AUTO launch took, 1 or 3 (launch_shape writes it; a diagnostic)
    uint32_t *rerun_seen;        // page-locked HOST word the device can write (the instance's): how many channels the rerun pass of the
                                 // last call found on its list -- written by that pass, read by the host WITHOUT synchronising when it sizes
                                 // the next rerun pass (a stale value only picks the other grid: results do not depend on it)
    AgcParams agcp;
};

// words of RxParams::flags
enum { kFlagNanInf = 0, kFlagWords = 4 };
// RxParams::rerun_flag words
enum : uint32_t { kFlagRerun = 1u, kProvShift = 1u, kProvMask = 3u, kProvExact = 0u, kProvSplitExt = 1u, kProvSplit = 2u, kExtBufShift = 3u,
                  kExtQ15 = 16u,       // bit 4: the hist_ext row holds the RAW int16 samples of an int16-slot call (half the bytes; k_hist_exact mixes them again)
                  kFlagHold = 32u,     // bit 5: held on the exact kernel (hysteresis of SELENITE_ARITH_AUTO)
                  kFlagClean1 = 64u,   // bit 6: ... and its last call there had no block near the guard ratio (the second such call in a row hands it back)
                  kLvlMask = 0xFFFFFF00u };

__host__ __device__ inline bool mode_is_cw(uint32_t m) { return m == SELENITE_MODE_CW || m == SELENITE_MODE_CWR; }
__host__ __device__ inline bool mode_is_upper(uint32_t m)
{
    return m == SELENITE_MODE_USB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_CW;
}

// ---- generic path (rx_generic.hip): any configuration, three simple kernels ----
// front: NCO -> decimator -> demodulator, un-scaled audio (f32) to `audio`
hipError_t launch_front_generic(const RxParams &p, int arith, const void *src, bool src_q15,
                                float *audio, hipStream_t st);
size_t front_generic_lds_bytes(const RxParams &p);
// arm_q15_to_float over a whole buffer (SupportFunctions/arm_q15_to_float.c:87): n int16 values -> n floats
hipError_t launch_q15_to_f32(const int16_t *src, float *dst, size_t n, hipStream_t st);
// SELENITE_ARITH_AUTO, in front of the rerun pass: the Hilbert-pair history of every flagged channel whose state the call before left
// on the matrix kernel (with its hist_ext) is recomputed in exact arithmetic from hist_ext + the decimator state (rx_generic.hip)
// (all: a call that runs the exact kernel on every channel -- FM, a shape without a matrix kernel for this call length -- repairs every such channel)
hipError_t launch_hist_exact(const RxParams &p, bool all, hipStream_t st);
// CW biquad cascade, in place on f32 audio
hipError_t launch_biquad_generic(const RxParams &p, int arith, float *audio, hipStream_t st);
// per-channel AGC (or plain copy/convert when p.agc == 0): audio -> dst
hipError_t launch_agc_generic(const RxParams &p, int arith, const float *audio, void *dst,
                              bool dst_q15, hipStream_t st);
// global-gain AGC pieces: env[b] = max over channels of max|audio| in DSP block b
uint32_t env_global_rows(const RxParams &p);
hipError_t launch_env_global(const RxParams &p, const float *audio, float *part, float *env, hipStream_t st);
// env[b] = max over `rows` rows of part[row][b]; part holds env_fold_scratch_floats(rows, nblk) floats (scratch behind the rows)
size_t env_fold_scratch_floats(uint32_t rows, uint32_t nblk);
hipError_t launch_env_fold(float *part, float *env, uint32_t rows, uint32_t nblk, hipStream_t st);
hipError_t launch_agc_apply_global(const RxParams &p, int arith, const float *audio, const float *env,
                                   void *dst, bool dst_q15, hipStream_t st);

// ---- fused fast paths (rx_fused.hip); return false when the configuration is not covered ----
struct FusedPlan {
    int kind = 0;                 // 0 = none
    const char *name = "generic";
    std::string name_buf;         // storage behind `name` for the composed kernel names
    float *d_cq = nullptr;        // zero-padded decimator taps in the fused kernel's indexing
    float *d_btab = nullptr;      // banded-Toeplitz B operand of the MFMA decimator
    void *d_btab16 = nullptr;     // same operand split into f16 hi / lo parts (SELENITE_ARITH_SPLIT16)
    float split_post = 1.0f;      // 2^-(sample scale + tap scale) applied to the split-precision result (k_hilb_split16)
    int split_sc = 0;             // tap scale exponent of the split-precision decimator (k_ssb_split16)
    bool use_mfma = false;        // FMA arithmetic: decimator on the matrix cores
    bool dense = false;           // the DENSE flavour of k_ssb_fused: a FIR pair with arbitrary taps (kind = an ID of SRX_DENSE_SHAPES)
    float *d_ptab = nullptr;      // ... its tap tables [2][DenseTab::LEN]
    uint32_t dense_t0 = 0;        // ... first FIR step with a tap that is not padding
    bool dense_delay_impulse = false;   // ... the delay FIR is a unit impulse (only the Hilbert FIR runs dense)
    bool tables_built = false;
};
hipError_t plan_fused(const selenite_rx_config &cfg, bool delay_is_impulse, int delay_index,
                      bool hilb_odd_only, FusedPlan &plan);
void free_fused(FusedPlan &plan);
// passes k_ssb_split16 runs: 256 audio samples, or fewer when the DSP block does not divide 256 -- whole 16-output tiles
inline bool split16_pass_ok(uint32_t pass_out) { return pass_out != 0 && pass_out <= 256u && pass_out % 16u == 0; }
// audio samples a full pass of k_ssb_split16 produces: the largest whole number of DSP blocks in its tile -- 256 outputs, or 128 when the
// chain decimates by 8 (the by-4 product with every second output kept: FusedArgs::dec2)
inline uint32_t split16_pass_out(uint32_t block, uint32_t decim)
{
    const uint32_t na = decim ? block / decim : 0u;
    return na ? (decim == 8u ? 128u : 256u) / na * na : 0u;
}
bool fused_tail_split(const FusedPlan &plan, const selenite_rx_config &cfg, uint32_t block_size);
hipError_t launch_fused(const FusedPlan &plan, const RxParams &p, int arith, const void *src,
                        bool src_q15, void *dst, bool dst_q15, int delay_index, hipStream_t st);

// the instantiated decimator length that serves an instance of nd taps (its taps zero-padded in front): fused kernels / k_ssb_split16; -1: none
int fused_template_nd(int nd, int m, int nh);
int split16_template_nd(int nd, int m, int nh);
// true when k_ssb_split16 of this shape has the periodic-LO flavour (a pass is a whole number of 256-sample periods)
bool ssb_split16_periodic_lo(int nd, int m, int nh);
// true when rx_split16.hip instantiates k_ssb_split16 for this shape
bool ssb_split16_has_shape(int nd, int m, int nh);

// fused CW kernel (rx_cw.hip): NCO -> real part -> 4-stage biquad cascade -> AGC
bool cw_fused_ok(const selenite_rx_config &cfg, uint32_t block_size);
bool cw_strides_ok(uint64_t in_stride, uint64_t out_stride);
// no-DSP kernel with the fetch pattern, launch shape and residency of k_cw_fused (rx_cw.hip; bench.py `pattern_roof`)
hipError_t launch_cw_roof(const RxParams &p, const void *src, bool q15, void *dst, float4 *state, uint32_t work, hipStream_t st);
hipError_t launch_cw_fused(const RxParams &p, const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st);

// shared local-oscillator table for one call (rx_fused.hip)
hipError_t launch_lo_table(float2 *lo, const float *sintab, uint32_t phase0, uint32_t step, uint32_t nsamp,
                           hipStream_t st);

// ---- synthetic input (rx_synth.hip) ----
hipError_t launch_synth(float *dIQ, const float *sintab, uint32_t first_channel, uint32_t nch,
                        uint64_t first_sample, uint32_t nsamp, uint64_t seed, hipStream_t st);
void synth_host(float *iq, const float *sintab, uint32_t first_channel, uint32_t nch,
                uint64_t first_sample, uint32_t nsamp, uint64_t seed);

// no-arithmetic streaming kernel with the traffic of one process call (rx_synth.hip; bench.py `streaming_roof`)
hipError_t launch_stream_roof(const void *in, void *out, float *state, uint32_t channels, uint32_t in_bytes, uint32_t out_bytes,
                              uint32_t state_words, hipStream_t st);

// host copy of sinTable_f32 regenerated from its documented generator (rx_api.hip)
const float *host_sin_table();

}  // namespace srx

struct selenite_rx_instance {
    selenite_rx_config cfg;            // pointers inside refer to the host copies below
    std::vector<float> h_dec, h_hilb, h_delay, h_biq;
    std::vector<uint32_t> h_step;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    float *d_dec_c = nullptr, *d_hilb_c = nullptr, *d_delay_c = nullptr, *d_biq_c = nullptr, *d_sintab = nullptr;
    uint32_t *d_step = nullptr, *d_phase = nullptr;
    float *d_dec_state = nullptr, *d_fir_state = nullptr, *d_biq_state = nullptr, *d_gain = nullptr;
    uint32_t *d_flags = nullptr;       // kFlagWords words
    uint32_t *d_guard_ch = nullptr;    // [3][channels] guarded DSP blocks per channel | process calls with a guarded block | handover blocks (sticky)
    uint32_t *d_rerun_flag = nullptr;  // [channels] SELENITE_ARITH_AUTO: rerun bit + state provenance (RxParams::rerun_flag)
    uint32_t *d_rerun_list = nullptr;  // [channels + 2] dense list of the channels to recompute, behind its two alternating counters (RxParams::chan_list)
    uint32_t rerun_par = 0;            // which counter the next call counts in
    uint32_t *h_rerun_seen = nullptr;  // page-locked, device-writable: entries of the last rerun pass's list (RxParams::rerun_seen)
    int auto_launches = 1;             // selenite_rx_set_auto_launches: 1 = one launch where the matrix kernel recomputes a channel itself, 3 = never
    uint32_t auto_form_last = 0;       // 1 / 3: the form of the last SELENITE_ARITH_AUTO launch on a matrix kernel (selenite_rx_auto_launches_last)
    float2 *d_hist_ext = nullptr;      // [2][channels][ext_len] SELENITE_ARITH_AUTO with k_ssb_split16: RxParams::hist_ext
    uint32_t ext_len = 0;
    bool handover_repair = true;       // selenite_rx_set_handover_repair
    float guard_ratio = 0.25f;         // -12 dB
    bool steps_grid256 = false;        // every NCO step is a multiple of 2^24: every channel's LO repeats every 256 samples
    float *d_scratch = nullptr;  size_t scratch_bytes = 0;   // intermediate f32 audio
    float *d_conv_in = nullptr;  size_t conv_in_bytes = 0;   // f32 copy of int16 input (int16 slots with a global gain on the fused kernels)
    float *d_env = nullptr;      size_t env_cap = 0;
    float *d_env_part = nullptr; size_t env_part_cap = 0;   // per-wavefront envelope maxima
    void *d_io_in = nullptr;     size_t io_in_bytes = 0;     // staging for the host-pointer entry points (global-gain calls)
    void *d_io_out = nullptr;    size_t io_out_bytes = 0;
    // chunked, double-buffered pipeline of the host-pointer entry points (rx_api.hip: process_host)
    struct HostPipe {
        hipStream_t h2d = nullptr, d2h = nullptr;
        hipEvent_t ev_in[2] = { nullptr, nullptr }, ev_done[2] = { nullptr, nullptr }, ev_out[2] = { nullptr, nullptr };
        void *d_in[2] = { nullptr, nullptr }, *d_out[2] = { nullptr, nullptr };     // device chunk buffers
        void *h_in[2] = { nullptr, nullptr }, *h_out[2] = { nullptr, nullptr };     // pinned staging (pageable callers only)
        size_t d_in_bytes = 0, d_out_bytes = 0, h_in_bytes = 0, h_out_bytes = 0;
    } pipe;
    uint32_t sub_first = 0, sub_count = 0;                   // channel sub-range of the current launch (0 = all channels)
    float2 *d_lo = nullptr;      size_t lo_bytes = 0;        // shared LO table of the current call
    bool lo_valid = false; uint32_t lo_phase = 0, lo_step = 0, lo_n = 0;   // what d_lo holds: LO[n], n < lo_n, from (lo_phase, lo_step)
    bool steps_uniform = false;        // every channel has the same NCO step
    bool phase_uniform = true;         // ... and the same phase (true after init/reset)
    uint32_t phase_host = 0;           // that common phase, tracked on the host
    bool delay_is_impulse = false; int delay_index = 0; bool hilb_odd_only = false;
    srx::FusedPlan plan;
    int no_shared_lo = 0;              // SELENITE_RX_NO_SHARED_LO=1: always compute the LO per channel
    int no_periodic_lo = 0;            // SELENITE_RX_NO_PERIODIC_LO=1: never keep a periodic shared LO in registers
    int force_generic = 0;             // SELENITE_RX_FORCE_GENERIC=1 (tests cross-check both paths)
    int status = SELENITE_RX_SUCCESS;
    std::string err;
};
