// rx_cw.hip -- fused CW narrow-filter kernel (BASELINE cfg4): NCO/BFO mix -> real part ->
// arm_biquad_cascade_df1_f32 (NS stages) -> AGC, one launch, gfx950.
//
// The DF1 recurrence (arm_biquad_cascade_df1_f32.c:220) has no time parallelism without changing
// the rounding, so parallelism is channels x stages.  A wavefront is a SYSTOLIC array:
//
//      lane = 16*row + CPR*stage + channel-in-row      (NS = 2, 4 or 8 stages; CPR = 16 / NS channels per 16-lane DPP row:
//                                                       32, 16 or 8 channels per wavefront, stage-major inside a row)
//
// at step k stage s works on sample k-s; a stage's output reaches the next stage's lane with one
// DPP row_shr:CPR.  Stage-major rows put the stage-0 lanes of a row into whole DPP banks (2 and 4 stages), so the
// choice "stage 0 takes the input sample, the others their neighbour's output" is the bank mask of that one DPP move.  Value for
// value this is the reference's stage-outer loop: each stage consumes exactly the previous stage's
// output sequence, left-to-right sums, feedback added, no fusion (both arithmetic modes -- the
// recurrence keeps the reference rounding).
//
// Per DSP block (BLK samples) and workgroup (one wavefront, 16 channels with 4 stages):
//   1. buffer loads (2 complex samples per lane, 1 KB of one channel per instruction with 4 stages), NCO mix real part -- into
//      REGISTERS first, then into ONE LDS tile x[16][BLK+4] (f32).  Shared LO: all loads of a chunk use the same LO float4.
//   2. BLK+NS-1 systolic steps (NS-1 masked fill + NS-1 masked drain steps so a block's envelope is complete before it is scaled), 8 vector
//      instructions each (9 with 8 stages); the last stage writes y[n] over x[n] in place (x[n] was consumed NS-1 steps ago) and tracks max|y|.
//   3. AGC gain law per channel; the tile goes out as 16 rows of 1 KiB: ds_read_b128, scale by the channel's gain (v_readlane), one
//      non-temporal buffer store per lane per row.
// State (4 floats per channel-stage) is one coalesced dwordx4 load/store per lane per call.
// What bounds it (profiles/r6/cfg4_pattern_roof.md): NOT its fetch pattern (sixteen 1 KB pieces 32 KB apart per burst: a no-DSP kernel with exactly
// these bursts is faster than one with 4 KB runs) but the balance of ~1360 vector instructions per chunk of a wave against the memory time of a
// chunk, with two waves per SIMD to overlap them.  Hence: two chunks in flight (register slot q & 1), as few vector instructions per step as
// the data flow allows (the step is bound by their COUNT: 4 cycles each at the SIMD whatever the number of waves), workgroups persistent over
// channel groups.  Carried over from round 5: a wait for loaded data is a wait for every vector-memory operation in flight whenever stores are
// (s_waitcnt vmcnt(0)), so the first chunk of a block is taken out of its load registers BEFORE the 16 KB store burst of the block in front of it
// goes out; the delay lines are (x1, y1) / (x2, y2) register pairs that swap names.
#include "rx_internal.h"
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace srx {

__device__ __forceinline__ void cw_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float cw_v2 __attribute__((ext_vector_type(2)));

// four consecutive stage-0 inputs of a channel row, as FOUR dword reads (volatile: not merged into one 16-byte read): each lands in a
// register of its own, and the step's DPP move turns it in place into the x half of an (x, y) delay-line pair -- out of an aligned quad
// it had to be copied there, one vector move per step
struct cw_x4 { float x, y, z, w; };
__device__ __forceinline__ cw_x4 cw_ld4(const float *p)
{
    typedef __attribute__((address_space(3))) float lds_f32;
    const volatile lds_f32 *q = (const volatile lds_f32 *)p;
    cw_x4 r;
    r.x = q[0]; r.y = q[1]; r.z = q[2]; r.w = q[3];
    return r;
}
typedef unsigned int u4v_cw __attribute__((ext_vector_type(4)));
typedef unsigned int u2v_cw __attribute__((ext_vector_type(2)));

// the input of a lane's stage at one step: the sample `xs` in the stage-0 lanes, elsewhere the output `y` of the lane CPR below (the
// stage before it, same channel) -- lanes of a row are stage-major, so the stage-0 lanes are banks 0 .. CPR/4-1 of the row
template <int NS>
__device__ __forceinline__ float cw_stage_input(float xs, float y, bool first)
{
    constexpr int CPR = 16 / NS;
    if constexpr (CPR >= 4) {
        // ONE DPP move: row_shr:CPR into the banks of stages 1 .. NS-1; the masked-out banks keep `old` = xs (xs dies here: no copy)
        constexpr int bank_mask = 0xf & ~((1 << (CPR / 4)) - 1);
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xs), __float_as_int(y), 0x110 + CPR, 0xf, bank_mask, false));
    } else {
        // 8 stages: the stage-0 lanes are half a bank -- shift (they read 0: bound_ctrl), then select
        const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x110 + CPR, 0xf, 0xf, true));
        return first ? xs : prev;
    }
}

// two complex samples per lane and load, through a buffer descriptor over the workgroup's channels: the lane's part of the address is
// one 32-bit offset for the whole call (no 64-bit address arithmetic per load), and lanes past the last channel read zeros
template <typename T> struct CwRaw;
template <> struct CwRaw<float> {
    typedef u4v_cw type;
    static constexpr int kBytes = 16;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2); }      // nt: streamed once
    static __device__ __forceinline__ void unpack(const type &r, cw_v2 &a, cw_v2 &b)
    {
        a = cw_v2{ __uint_as_float(r.x), __uint_as_float(r.y) }; b = cw_v2{ __uint_as_float(r.z), __uint_as_float(r.w) };
    }
};
template <> struct CwRaw<int16_t> {
    typedef u2v_cw type;
    static constexpr int kBytes = 8;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2); }
    static __device__ __forceinline__ void unpack(const type &r, cw_v2 &a, cw_v2 &b)
    {
        a = cw_v2{ q15_to_float((int16_t)(r.x & 0xffffu)), q15_to_float((int16_t)(r.x >> 16)) };
        b = cw_v2{ q15_to_float((int16_t)(r.y & 0xffffu)), q15_to_float((int16_t)(r.y >> 16)) };
    }
};

// Geometry of the systolic array for NS stages per channel (NS = 2, 4, 8: lanes of a channel are adjacent and never
// straddle a 16-lane DPP row): CH = 64 / NS channels per wavefront, input chunks of CS = 1024 / CH samples (eight
// dwordx4 loads of two complex samples per lane and chunk), D = NS - 1 fill / drain steps per DSP block.  Stage NS-1
// emits y[k - D] at step k; trips are four steps, so an aligned float4 of outputs is NCAR values carried from the
// trip before plus the first 4 - NCAR of this one (NCAR = 4 - D mod 4), PRO = D / 4 + 1 trips behind the input.
template <int NS, int NL = 8>
struct CwGeo {
    static_assert(NS == 2 || NS == 4 || NS == 8, "systolic CW kernel: 2, 4 or 8 biquad stages");
    static constexpr int CH = 64 / NS;
    static constexpr int CS = NL * 128 / CH;        // samples per input chunk: NL wave loads of two complex samples per lane
    static constexpr int LPC = CS / 2;              // lanes per channel row of a load
    static constexpr int CPL = 64 / LPC;            // channels per load
    static constexpr int D = NS - 1;
    static constexpr int R = D % 4;                 // outputs of a trip that complete the pending group
    static constexpr int NCAR = 4 - R;
    static constexpr int PRO = D / 4 + 1;
    static_assert(LPC <= 64 && 64 % LPC == 0, "a load covers whole channel rows");
};
// bursts in flight (k_cw_fused and its pattern roof): two where a DSP block is an even number of chunks and the registers allow it
template <int NCHUNK, int NCO, typename TIn> struct CwDepth { static constexpr int value = (NCHUNK % 2 == 0 && !(NCO == 1 && sizeof(TIn) == 4)) ? 2 : 1; };
// loads per chunk:
// 16 where the DSP block is whole chunks of that size and a chunk row still fits a wave load (2 / 4 stages): 1 KB of ONE channel per load
// instruction with 4 stages, half as many chunk prologues; else 8
template <int NS, int BLK> struct CwLoads { static constexpr int NL = (NS <= 4 && BLK % (2048 / (64 / NS)) == 0) ? 16 : 8; };

template <int NS, int NCO, int BLK, typename TIn, typename TOut>
__global__ __launch_bounds__(64) void k_cw_fused(RxParams p, const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    constexpr int NL = CwLoads<NS, BLK>::NL;
    using CG = CwGeo<NS, NL>;
    constexpr int CH = CG::CH, CS = CG::CS, D = CG::D, R = CG::R, NCAR = CG::NCAR, PRO = CG::PRO;
    constexpr int RS = BLK + 4;                         // tile row stride (floats); rows stay 16 B aligned
    constexpr int NCHUNK = BLK / CS;                    // input chunks of CS samples x CH channels
    constexpr int TPC = CS / 4;                         // trips per chunk
    static_assert(BLK % CS == 0 && PRO <= TPC && 4 * PRO <= BLK, "block holds whole chunks; prologue inside the first chunk");
    __shared__ __attribute__((aligned(16))) float tile[CH * RS];
    __shared__ float tab[NCO == 1 ? 516 : 4];
    constexpr int CPR = 16 / NS;                        // channels per 16-lane row
    const int lane = threadIdx.x;
    const int s = (lane & 15) / CPR, ch = (lane >> 4) * CPR + (lane & (CPR - 1));
    // Persistent over channel groups: workgroup b takes the groups b, b + gridDim.x, ... (the launcher hands out whole multiples only, else one
    // group per workgroup) and requests the first two chunks of its NEXT group -- and that group's state -- during the last DSP block of the
    // present one: a group switch costs no cold memory round trip and no workgroup launch.
    const uint32_t ngroups = (p.channels + CH - 1) / CH;
    uint32_t grp = blockIdx.x;
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // per-lane stage constants and state
    const float b0 = p.biq_c[5 * s], b1 = p.biq_c[5 * s + 1], b2 = p.biq_c[5 * s + 2];
    const float a1 = p.biq_c[5 * s + 3], a2 = p.biq_c[5 * s + 4];
    bool nonfinite = false;                                           // any audio sample of this wavefront NaN / Inf
    // the last workgroup of a channel count that is not a multiple of CH: lanes past the end work on a
    // copy of the last channel (every load index is clamped) and none of their stores is issued
    auto chan_of = [&](uint32_t g) { const uint32_t cc = g * CH + ch; return cc < p.channels ? cc : p.channels - 1; };
    struct GroupState { float4 st; float gain; uint32_t ph, stp; };
    auto load_group_state = [&](uint32_t g) {
        const uint32_t cc = chan_of(g);
        GroupState gs;
        gs.st = *reinterpret_cast<const float4 *>(p.biq_state + ((size_t)cc * NS + s) * 4);      // { x1, x2, y1, y2 } (arm_biquad_cascade_df1_f32.c:73-82)
        gs.gain = p.agc ? p.gain[cc] : 1.0f;
        gs.ph = NCO ? p.phase[cc] : 0u; gs.stp = NCO ? p.step[cc] : 0u;
        return gs;
    };
    GroupState gs_cur = load_group_state(grp);
    // load-phase geometry: load j of a chunk covers channel CPL*j + lane/LPC, samples 2*(lane%LPC), +1
    const int lch = lane / CG::LPC, lsm = 2 * (lane % CG::LPC);
    // last-stage lanes write y[4i..4i+3] over x[4i..4i+3] of their channel's row (four dwords from wherever the step left them -- a 16-byte
    // store would want them copied into one aligned register quad first, and instructions are what this kernel is short of)
    const bool last = s == NS - 1;
    float *wrow = tile + ch * RS;
    const float *rbase = tile + ch * RS;
    typedef __attribute__((address_space(3))) float lds_f32;
    const uint32_t wbase = (uint32_t)(uintptr_t)(lds_f32 *)wrow;      // LDS byte address of the channel row (output groups: 16 bytes per trip)
    constexpr uint32_t winc = 16u;
    const uint64_t last_mask = __builtin_amdgcn_ballot_w64(last);     // the last-stage lanes (wave-uniform: an SGPR pair)

    typedef typename CwRaw<TIn>::type raw_t;
    // input prefetch: TWO chunks ahead where a DSP block is an even number of chunks (chunk q of every block lives in register slot q & 1:
    // 2 x NL wave loads in flight), else one.  The systolic steps of a chunk are ~1500 dependent vector instructions, about as long as the
    // loaded memory system takes to deliver a burst (profiles/r6/cfg4_pattern_roof.txt: a no-arithmetic kernel with this fetch pattern and
    // that many instructions between its bursts runs 9 % faster with two bursts in flight than with one; the pattern itself -- sixteen
    // 1 KB pieces 32 KB apart -- costs nothing against 4 KB runs).  (Round 5's "two chunks of 8 loads ahead" kept the same 16 KB in flight.)
    // (per-channel arm_sin/cos with f32 slots: its registers do not leave room for a second slot at two waves per SIMD)
    constexpr int DEPTH = CwDepth<NCHUNK, NCO, TIn>::value;
    raw_t raw[DEPTH][NL];
    u4v_cw lo4[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) lo4[d] = u4v_cw{ 0u, 0u, 0u, 0u };
    // the workgroup's CH channels (fewer in the last workgroup: the range ends with the array), 32-bit offsets inside them
    constexpr int EB = CwRaw<TIn>::kBytes / 2;                        // bytes per complex sample
    auto in_rsrc = [&](uint32_t g) {                                   // the CH channel rows of group g (none past the last group: every load reads 0, no access)
        const uint32_t cg = g * CH, n = g < ngroups ? min((uint32_t)CH, p.channels - cg) : 0u;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<TIn *>(src) + (size_t)(n ? cg : 0u) * p.in_stride * 2, 0,
                                                 n ? (int)(((n - 1) * p.in_stride + p.block_size) * EB) : 0, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rs_in = in_rsrc(grp);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(p.lo), 0, NCO == 2 ? (int)(p.block_size * 8u) : 0, 0x00020000);
    const int voff_in = (int)((lch * p.in_stride + lsm) * EB);        // this lane's part of every load address
    const int joff_in = (int)(CG::CPL * p.in_stride * EB);            // ... load j of a chunk adds j of these (wave-uniform)
    auto issue_loads = [&](auto slot, const __amdgpu_buffer_rsrc_t &rs, uint32_t n_first) {          // chunk of CS samples x CH channels of the group behind `rs` starting at n_first, into register slot `slot`
        constexpr int SL = decltype(slot)::value;
#pragma unroll
        for (int j = 0; j < NL; ++j) raw[SL][j] = CwRaw<TIn>::load(rs, voff_in, (int)(n_first * EB) + j * joff_in);
        if constexpr (NCO == 2) lo4[SL] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, lsm * 8, (int)(n_first * 8u), 0);
    };
    float *mrow = tile + lch * RS + lsm;                 // this lane's slot of load 0 in chunk 0
    // The chunk's samples are mixed into REGISTERS (mix_regs: this is where the wave waits for its loads) and written to the tile later
    // (tile_write).  In between sits, at a block boundary, the audio store burst of the block before: loads and stores complete out of
    // order with respect to each other on this target, so a wait for loaded data is a wait for every vector-memory operation in flight
    // (s_waitcnt vmcnt(0)) -- behind the burst that was a wait for 16 KB of write acknowledgements per block.
    uint32_t c0 = grp * CH;                              // first channel of the present group
    float xm[NL][2];
    auto mix_regs = [&](auto slot, uint32_t n_first) {  // NCO mix (real part) of the chunk loaded into `slot`
        constexpr int SL = decltype(slot)::value;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            cw_v2 a, b;
            CwRaw<TIn>::unpack(raw[SL][j], a, b);
            float xa, xb;
            if constexpr (NCO == 0) {
                // (a move, so that the load registers are free for the next request: left as an alias the loads of the slot's next chunk
                // get registers of their own -- three chunks of them)
                asm("v_mov_b32 %0, %1" : "=v"(xa) : "v"(a.x));
                asm("v_mov_b32 %0, %1" : "=v"(xb) : "v"(b.x));
            } else {
                cw_v2 la, lb;
                if constexpr (NCO == 2) {
                    la = cw_v2{ __uint_as_float(lo4[SL].x), __uint_as_float(lo4[SL].y) }; lb = cw_v2{ __uint_as_float(lo4[SL].z), __uint_as_float(lo4[SL].w) };
                } else {
                    const uint32_t cj = min(c0 + CG::CPL * j + lch, p.channels - 1);
                    const uint32_t phj = p.phase[cj], stj = p.step[cj];
                    lo_v2f va, vb;                                    // arm_sin/cos_f32 restated for the vector ALU: same bits (rx_device.h)
                    const uint32_t pe = phj + (n_first + lsm) * stj;
                    nco_lo_pair(tab, pe, pe + stj, va, vb);
                    la = cw_v2{ va.x, va.y }; lb = cw_v2{ vb.x, vb.y };
                }
                // arm_cmplx_mult_cmplx_f32 real part, a c - b d: both products of a sample in one packed multiply, rounded, then the difference
                // (the differences through asm: left to itself the vectorizer gathers them into one packed subtract behind four register copies)
                const cw_v2 ta = a * la, tb = b * lb;
                asm("v_sub_f32 %0, %1, %2" : "=v"(xa) : "v"(ta.x), "v"(ta.y));
                asm("v_sub_f32 %0, %1, %2" : "=v"(xb) : "v"(tb.x), "v"(tb.y));
            }
            xm[j][0] = xa; xm[j][1] = xb;
        }
    };
    auto tile_write = [&](int q) {
#pragma unroll
        for (int j = 0; j < NL; ++j) *reinterpret_cast<float2 *>(mrow + CG::CPL * j * RS + CS * q) = make_float2(xm[j][0], xm[j][1]);
    };
    // One DF1 step of this lane's stage; xs = stage-0 input of the step.  The delay lines live in TWO register pairs, A = (x1, y1) and
    // B = (x2, y2): the four products of a step are two packed multiplies, A * (b1, a1) and B * (b2, a2), and the shift of both delay
    // lines is a change of NAMES -- B's registers are dead once its products are formed, so the step writes the new (x1, y1) = (xin, y)
    // into them and the next step calls the pairs the other way round.  No register copy, no operand swizzle: 8 vector instructions per
    // step (DPP move that is also the input select, 1 + 2 multiplies, 4 adds; 9 with 8 stages).  Same products, same left-to-right sum as the reference
    // (arm_biquad_cascade_df1_f32.c:220: b0 x + b1 x1 + b2 x2 + a1 y1 + a2 y2).
    typedef float cw_v2f __attribute__((ext_vector_type(2)));
    const cw_v2f c1 = { b1, a1 }, c2 = { b2, a2 };
    cw_v2f PA, PB;                                                    // (x1, y1), (x2, y2): set when a group starts
    auto step = [&](float xs, cw_v2f &A, cw_v2f &B) -> float {        // on return B holds the new (x1, y1), A the new (x2, y2)
        const float xin = cw_stage_input<NS>(xs, A.y, s == 0);
        const cw_v2f t1 = A * c1, t2 = B * c2;
        const float p0 = b0 * xin;
        float y = p0 + t1.x;
        y = y + t2.x;
        y = y + t1.y;
        y = y + t2.y;
        B = cw_v2f{ xin, y };
        return y;
    };
    // four steps of the steady state: an even number, so PA is (x1, y1) again behind them
    auto trip4 = [&](const cw_x4 &xq, float (&o)[4]) {
        o[0] = step(xq.x, PA, PB);
        o[1] = step(xq.y, PB, PA);
        o[2] = step(xq.z, PA, PB);
        o[3] = step(xq.w, PB, PA);
    };
    // one step with the state update suppressed where stage s has no sample at step k (fill / drain); PA stays (x1, y1)
    auto step_masked = [&](float xs, int k) -> float {
        const cw_v2f oa = PA, ob = PB;
        const float y = step(xs, PA, PB);                             // PB = new (x1, y1), PA = old (x1, y1) = new (x2, y2)
        const bool valid = (k - s >= 0) && (k - s < BLK);
        const cw_v2f n1 = PB;
        PA.x = valid ? n1.x : oa.x; PA.y = valid ? n1.y : oa.y;
        PB.x = valid ? oa.x : ob.x; PB.y = valid ? oa.y : ob.y;
        return y;
    };

    const uint32_t nblk = p.block_size / BLK;
    typedef std::integral_constant<int, 0> slot0_t;
    typedef std::integral_constant<int, DEPTH - 1> slot1_t;
    __amdgpu_buffer_rsrc_t rs_next = in_rsrc(grp + gridDim.x);
    auto mix_and_prefetch = [&](auto slot, uint32_t n_first) {     // the loaded chunk (at n_first) out of its registers, the chunk DEPTH behind it requested into them
        mix_regs(slot, n_first);
        const uint32_t nxt = n_first + DEPTH * CS;         // (may be in the next block -- or in the workgroup's next group: its first chunks)
        if (nxt < p.block_size) issue_loads(slot, rs_in, nxt);
        else if constexpr (DEPTH == 2) issue_loads(slot, rs_next, nxt - p.block_size);
    };
    if constexpr (DEPTH == 2) { issue_loads(slot0_t{}, rs_in, 0); issue_loads(slot1_t{}, rs_in, CS); }   // (a call is whole DSP blocks: at least two chunks)
    for (;;) {
    const bool live = c0 + ch < p.channels;
    const uint32_t c = chan_of(grp);
    const uint32_t chs = min((uint32_t)CH, p.channels - c0);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)c0 * p.out_stride, 0,
                                                                            (int)(((chs - 1) * p.out_stride + p.block_size) * (uint32_t)sizeof(TOut)), 0x00020000);
    PA = cw_v2f{ gs_cur.st.x, gs_cur.st.z }; PB = cw_v2f{ gs_cur.st.y, gs_cur.st.w };
    float gain = gs_cur.gain;
    const uint32_t ph_own = gs_cur.ph, st_own = gs_cur.stp;
    const uint32_t grp_next = grp + gridDim.x;
    if constexpr (DEPTH == 1) issue_loads(slot0_t{}, rs_in, 0);     // (one slot: nothing of this group was requested ahead)
    mix_and_prefetch(slot0_t{}, 0);
    cw_lds_sync();
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t n0 = blk * BLK;
        float m = 0.0f;
        float car[NCAR];                                              // outputs waiting for the rest of their float4
#pragma unroll
        for (int v = 0; v < NCAR; ++v) car[v] = 0.0f;
        auto chunk = [&](int q, auto slot) {
            // ---- 1. this chunk's input into the tile; the HBM loads of the chunk that takes its register slot next go out meanwhile
            // (the first chunk of a block was mixed in front of the store burst of the block before it: below) ----
            if (q != 0) mix_and_prefetch(slot, n0 + CS * q);
            tile_write(q);
            cw_lds_sync();
            // ---- 2. TPC trips of 4 systolic steps; stage-0 input is one aligned float4 per trip ----
            int i = TPC * q;
            cw_x4 xq = cw_ld4(rbase + 4 * i);
            if (q == 0) {                                             // block prologue: the D fill steps, first outputs into `car`
#pragma unroll
                for (int tpro = 0; tpro < PRO; ++tpro) {
                    const cw_x4 xn = cw_ld4(rbase + 4 * (tpro + 1));
                    const float o[4] = { step_masked(xq.x, 4 * tpro), step_masked(xq.y, 4 * tpro + 1),
                                         step_masked(xq.z, 4 * tpro + 2), step_masked(xq.w, 4 * tpro + 3) };
                    if (tpro == PRO - 1) {
#pragma unroll
                        for (int j = R; j < 4; ++j) {                 // y[0 .. NCAR-1] (last-stage lanes)
                            car[j - R] = o[j];
                            m = fmaxf(m, fabsf(o[j]));
                        }
                    }
                    xq = xn;
                }
                i = PRO;
            }
            // one trip: four steps, the aligned group y[4(i-PRO) .. +3] -- NCAR outputs carried from the trip before, then the first R of
            // this one -- into the tile (last-stage lanes; four dword stores from the registers the steps left them in)
            auto do_trip = [&](uint32_t wa, auto sub, const cw_x4 &xin4) {       // wa: LDS address of the output group of the round's first trip; sub: which trip of the round
                float o[4];
                trip4(xin4, o);
                float gq[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) gq[v] = v < NCAR ? car[v] : o[v - NCAR];
                // the last-stage lanes write their four values into their channel's row as two ds_write2_b32 from the registers the steps left them
                // in (no copies into an aligned quad); the execution mask is narrowed to those lanes by two scalar moves around the pair -- the
                // wave is whole here -- instead of a compare / branch / restore around four masked stores
                {
                    constexpr int O = 4 * decltype(sub)::value;
                    asm volatile("s_mov_b64 exec, %9\n\tds_write2_b32 %0, %1, %2 offset0:%5 offset1:%6\n\tds_write2_b32 %0, %3, %4 offset0:%7 offset1:%8\n\ts_mov_b64 exec, -1"
                                 : : "v"(wa), "v"(gq[0]), "v"(gq[1]), "v"(gq[2]), "v"(gq[3]), "n"(O), "n"(O + 1), "n"(O + 2), "n"(O + 3), "s"(last_mask) : "memory");
                }
                m = fmaxf(fmaxf(m, fabsf(o[0])), fmaxf(fabsf(o[1]), fmaxf(fabsf(o[2]), fabsf(o[3]))));
#pragma unroll
                for (int v = 0; v < NCAR; ++v) car[v] = o[R + v];
            };
            // two trips per round: the input of trip i + 1 is read while trip i runs, that of trip i + 2 -- straight into the registers
            // trip i has just emptied -- while trip i + 1 runs (the read behind a chunk's last trip lands in the next chunk's columns or
            // in the row's slack: never used)
            const int iend = TPC * q + TPC;
#pragma unroll 1
            for (; i + 1 < iend; i += 2) {
                const uint32_t wa = wbase + winc * (uint32_t)(i - PRO);
                const cw_x4 xb = cw_ld4(rbase + 4 * (i + 1));
                do_trip(wa, std::integral_constant<int, 0>{}, xq);
                xq = cw_ld4(rbase + 4 * (i + 2));
                do_trip(wa, std::integral_constant<int, 1>{}, xb);
            }
            if (i < iend) do_trip(wbase + winc * (uint32_t)(i - PRO), std::integral_constant<int, 0>{}, xq);                             // (the first chunk of a block: an odd number of trips behind the prologue)
        };
        if constexpr (DEPTH == 2) {
#pragma unroll 1
            for (int q = 0; q < NCHUNK; q += 2) { chunk(q, slot0_t{}); chunk(q + 1, slot1_t{}); }
        } else {
#pragma unroll 1
            for (int q = 0; q < NCHUNK; ++q) chunk(q, slot0_t{});
        }
        // ---- drain: stages 1..NS-1 finish samples BLK-D .. BLK-1 ----
        {
            float seq[4 * PRO];
#pragma unroll
            for (int v = 0; v < NCAR; ++v) seq[v] = car[v];
#pragma unroll
            for (int w = 0; w < D; ++w) {
                seq[NCAR + w] = step_masked(0.0f, BLK + w);
                m = fmaxf(m, fabsf(seq[NCAR + w]));
            }
#pragma unroll
            for (int tpro = 0; tpro < PRO; ++tpro)
                if (last)
                    *reinterpret_cast<float4 *>(wrow + BLK - 4 * PRO + 4 * tpro) =
                        make_float4(seq[4 * tpro], seq[4 * tpro + 1], seq[4 * tpro + 2], seq[4 * tpro + 3]);
        }
        cw_lds_sync();
        // ---- 3. AGC gain law (last-stage lanes hold max|y| of their channel) and scaled store ----
        if (p.agc) gain = agc_update<0>(p.agcp, gain, m);
        if (blk + 1 < nblk) mix_and_prefetch(slot0_t{}, n0 + BLK);   // (chunk 0 of every block: slot 0)
#pragma unroll 4
        for (int r = 0; r < CH; ++r) {
            const float g = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gain), 16 * (r / CPR) + CPR * (NS - 1) + (r % CPR)));     // (r is wave-uniform: a scalar lane select)
#pragma unroll
            for (int h = 0; h < (BLK / 4 + 63) / 64; ++h) {
                const int t = 4 * (lane + 64 * h);
                float4 v = lds_ld4f(tile + r * RS + (t < BLK ? t : 0));
                v.x = v.x * g; v.y = v.y * g; v.z = v.z * g; v.w = v.w * g;
                {   // ARM_MATH_NANINF (arm_math.h:405): x * 0 is NaN iff x is not finite
                    const float z = __builtin_fmaf(v.w, 0.0f, __builtin_fmaf(v.z, 0.0f, __builtin_fmaf(v.y, 0.0f, v.x * 0.0f)));
                    nonfinite = nonfinite || (t < BLK && (uint32_t)r < chs && z != z);
                }
                // buffer stores: rows past the workgroup's last channel and lanes past the block fall outside the descriptor's range and are dropped.
                // Non-temporal (written once, never read by the chain) unless a gain pass reads the audio back -- two instructions that differ in
                // their cache-policy immediate (as `if (cached) *p = v; else __builtin_nontemporal_store(v, p)` the two stores were merged into ONE
                // plain store: the audio of every CW call went through the caches)
                const int voff = t < BLK ? t * (int)sizeof(TOut) : 0x40000000;
                const int soff = (int)((r * p.out_stride + n0) * (uint32_t)sizeof(TOut));
                if constexpr (sizeof(TOut) == 4) {
                    const u4v_cw u = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) };
                    if (p.out_cached) __builtin_amdgcn_raw_buffer_store_b128(u, rs_out, voff, soff, 0);       // global gain, phase 1: the gain pass reads it back
                    else __builtin_amdgcn_raw_buffer_store_b128(u, rs_out, voff, soff, 2);
                } else {
                    uint32_t w0, w1;
                    float4_to_q15(v.x, v.y, v.z, v.w, p.q15_round, w0, w1);
                    __builtin_amdgcn_raw_buffer_store_b64(u2v_cw{ w0, w1 }, rs_out, voff, soff, 2);
                }
            }
        }
        cw_lds_sync();
    }
    if (live) {
        *reinterpret_cast<float4 *>(p.biq_state + ((size_t)c * NS + s) * 4) = make_float4(PA.x, PB.x, PA.y, PB.y);
        if (s == NS - 1) {
            if (p.agc) p.gain[c] = gain;
            if constexpr (NCO != 0) p.phase[c] = ph_own + p.block_size * st_own;
        }
    }
    if (grp_next >= ngroups) break;
    grp = grp_next; c0 = grp * CH; gs_cur = load_group_state(grp);
    rs_in = rs_next; rs_next = in_rsrc(grp + gridDim.x);
    }
    if (nonfinite) p.flags[kFlagNanInf] = 1u;                         // read by selenite_rx_sync / the host-pointer calls
}

// DSP blocks the systolic kernel is instantiated for: whole input chunks of CS = 1024 / (64 / NS) samples (32 / 64 / 128 for
// 2 / 4 / 8 stages) and a tile of (64 / NS) x (BLK + 4) floats inside the 64 KB of static LDS -- 128, 192, 256, 512 where that
// holds, and the firmware's 96-frame slot (Core/Inc/dsp_if.h:69-73) for 2 stages
static bool cw_block_ok(uint32_t ns, uint32_t blk)
{
    if (ns == 2) return blk == 96 || blk == 128 || blk == 192 || blk == 256;
    if (ns == 4) return blk == 128 || blk == 192 || blk == 256 || blk == 512;
    if (ns == 8) return blk == 128 || blk == 256 || blk == 512;
    return false;
}

bool cw_fused_ok(const selenite_rx_config &g, uint32_t block_size)
{
    // (block_size: the kernel addresses its workgroup's channels -- up to 32 rows of the input -- with 32-bit byte offsets)
    return mode_is_cw(g.mode) && g.nd_taps == 0 && g.decim == 1 && g.nh_taps == 0 && cw_block_ok(g.n_biquad, g.block) &&
           block_size % g.block == 0 && block_size <= (1u << 22);
}

// ... and the caller's channel strides (samples): 32 rows of either buffer inside 2^31 bytes -- a call with wider strides is served by the
// generic kernels (the dispatcher asks; launch_cw_fused refuses what it was not asked about)
bool cw_strides_ok(uint64_t in_stride, uint64_t out_stride)
{
    return in_stride * 32u * 8u < (1ull << 31) && out_stride * 32u * 4u < (1ull << 31);
}

// One workgroup per group of CH channels -- or, where the groups are a whole multiple K of (nearly) what the device keeps resident, that many
// workgroups taking K groups each, the next group's first chunks requested ahead (k_cw_fused / k_cw_roof: persistent over groups).  Never an uneven
// share: a one-shot grid lets the dispatcher balance what does not divide.
static uint32_t cw_grid(uint32_t ngroups, int resident)
{
    if (const uint32_t forced = plan_option(SELENITE_RX_OPT_CW_GRID)) return forced < ngroups ? forced : ngroups;      // (the tests: uneven shares, a partial last group)
    if (resident > 0 && ngroups > (uint32_t)resident) {
        const uint32_t k = (ngroups + (uint32_t)resident - 1) / (uint32_t)resident;
        if (ngroups % k == 0 && (uint64_t)(ngroups / k) * 100u >= (uint64_t)resident * 85u) return ngroups / k;
    }
    return ngroups;
}
template <typename K>
static int cw_resident(K kernel, size_t dyn_lds)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, dyn_lds) == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess &&
        hipGetDeviceProperties(&prop, dev) == hipSuccess)
        return per_cu * prop.multiProcessorCount;
    return -1;
}

template <int NS, int NCO, typename TIn, typename TOut>
static hipError_t cw_launch(const RxParams &p, const void *src, void *dst, hipStream_t st)
{
    constexpr int CH = CwGeo<NS>::CH;
    static const size_t pad = [] {                        // occupancy experiments (diagnostic builds): extra LDS per workgroup, inside what a workgroup may have
        const char *e = diag_env("SELENITE_RX_CW_LDS_PAD");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v > 0 && v <= 40 * 1024 ? v : 0);
    }();
    const uint32_t ngroups = (p.channels + CH - 1) / CH;
#define CW_BLK(B_)                                                                                                          \
    if (p.block == B_) {                                                                                                    \
        static int resident = 0;                                                                                            \
        if (resident == 0) resident = cw_resident(k_cw_fused<NS, NCO, B_, TIn, TOut>, pad);                                 \
        hipLaunchKernelGGL((k_cw_fused<NS, NCO, B_, TIn, TOut>), dim3(cw_grid(ngroups, resident)), dim3(64), pad, st, p,    \
                           static_cast<const TIn *>(src), static_cast<TOut *>(dst));                                       \
        return hipGetLastError();                                                                                           \
    }
    CW_BLK(256)
    CW_BLK(128)
    if constexpr (NS != 8) { CW_BLK(192) }
    if constexpr (NS != 2) { CW_BLK(512) }
    if constexpr (NS == 2) { CW_BLK(96) }
#undef CW_BLK
    return hipErrorNotSupported;
}

template <int NS>
static hipError_t cw_launch_ns(const RxParams &p, const void *src, bool q15, void *dst, hipStream_t st)
{
    if (q15) {
        if (p.nco == 2) return cw_launch<NS, 2, int16_t, int16_t>(p, src, dst, st);
        if (p.nco == 1) return cw_launch<NS, 1, int16_t, int16_t>(p, src, dst, st);
        return cw_launch<NS, 0, int16_t, int16_t>(p, src, dst, st);
    }
    if (p.nco == 2) return cw_launch<NS, 2, float, float>(p, src, dst, st);
    if (p.nco == 1) return cw_launch<NS, 1, float, float>(p, src, dst, st);
    return cw_launch<NS, 0, float, float>(p, src, dst, st);
}

hipError_t launch_cw_fused(const RxParams &p, const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st)
{
    if (src_q15 != dst_q15) return hipErrorNotSupported;
    if (!cw_strides_ok(p.in_stride, p.out_stride)) return hipErrorNotSupported;      // 32-bit offsets inside a workgroup's channels
    if (p.nbiq == 2) return cw_launch_ns<2>(p, src, src_q15, dst, st);
    if (p.nbiq == 4) return cw_launch_ns<4>(p, src, src_q15, dst, st);
    if (p.nbiq == 8) return cw_launch_ns<8>(p, src, src_q15, dst, st);
    return hipErrorNotSupported;
}

// ------------------------------------------------------------------------------------------
// k_cw_roof -- measurement support (bench.py `pattern_roof`, selenite_rx_time_pattern_roof_device): the memory traffic of k_cw_fused with
// ITS access pattern and launch shape and no DSP -- the same descriptors, the same bursts (NL wave loads per chunk: 1 KB of one channel row
// each with 4 stages, rows in_stride apart), the same number of bursts in flight, requested at the same points of the block loop, the
// 16 KB store burst of a DSP block behind the first chunk of the next, one float4 of state per lane in and out, one workgroup per CH
// channels.  One integer add per loaded register keeps the loads alive; `work` dependent vector instructions per chunk stand where the
// systolic steps are (0: the pattern alone).  What this takes is what ANY kernel with this fetch pattern takes.
// ------------------------------------------------------------------------------------------
template <int NS, int BLK, typename TIn, typename TOut>
__global__ __launch_bounds__(64) void k_cw_roof(RxParams p, const TIn *__restrict__ src, TOut *__restrict__ dst, float4 *__restrict__ state, uint32_t work)
{
    constexpr int NL = CwLoads<NS, BLK>::NL;
    using CG = CwGeo<NS, NL>;
    constexpr int CH = CG::CH, CS = CG::CS;
    constexpr int NCHUNK = BLK / CS;
    constexpr int DEPTH = CwDepth<NCHUNK, 2, TIn>::value;
    constexpr int EB = CwRaw<TIn>::kBytes / 2;
    extern __shared__ float roof_lds[];                      // (dynamic: sized by the launcher so that the residency is k_cw_fused's)
    const int lane = threadIdx.x;
    const uint32_t ngroups = (p.channels + CH - 1) / CH;
    auto in_rsrc = [&](uint32_t g) {
        const uint32_t cg = g * CH, n = g < ngroups ? min((uint32_t)CH, p.channels - cg) : 0u;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<TIn *>(src) + (size_t)(n ? cg : 0u) * p.in_stride * 2, 0,
                                                 n ? (int)(((n - 1) * p.in_stride + p.block_size) * EB) : 0, 0x00020000);
    };
    const int lch = lane / CG::LPC, lsm = 2 * (lane % CG::LPC);
    const int voff_in = (int)((lch * p.in_stride + lsm) * EB);
    const int joff_in = (int)(CG::CPL * p.in_stride * EB);
    typedef typename CwRaw<TIn>::type raw_t;
    raw_t raw[DEPTH][NL];
    uint32_t acc = 0u;
    auto issue = [&](auto slot, const __amdgpu_buffer_rsrc_t &rs, uint32_t n_first) {
        constexpr int SL = decltype(slot)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NL; ++j) raw[SL][j] = CwRaw<TIn>::load(rs, voff_in, (int)(n_first * EB) + j * joff_in);
    };
    auto consume = [&](auto slot) {
        constexpr int SL = decltype(slot)::value;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            acc += raw[SL][j].x + raw[SL][j].y;
            if constexpr (sizeof(raw_t) == 16) acc += raw[SL][j].z + raw[SL][j].w;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto busy = [&]() {
        float w0 = __uint_as_float(acc);
        for (uint32_t i = 0; i < work / 8u; ++i)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(w0));
        acc = __float_as_uint(w0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> slot0_t;
    typedef std::integral_constant<int, DEPTH - 1> slot1_t;
    const uint32_t nblk = p.block_size / BLK;
    uint32_t grp = blockIdx.x;
    __amdgpu_buffer_rsrc_t rs_in = in_rsrc(grp), rs_next = in_rsrc(grp + gridDim.x);
    auto request = [&](auto slot, uint32_t n_first) {                 // (past the call: the first chunks of the workgroup's next group, as k_cw_fused)
        if (n_first < p.block_size) issue(slot, rs_in, n_first);
        else if constexpr (DEPTH == 2) issue(slot, rs_next, n_first - p.block_size);
    };
    if constexpr (DEPTH == 2) { issue(slot0_t{}, rs_in, 0); issue(slot1_t{}, rs_in, CS); }
    for (;;) {
        const uint32_t c0 = grp * CH, chs = min((uint32_t)CH, p.channels - c0);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)c0 * p.out_stride, 0,
                                                                                (int)(((chs - 1) * p.out_stride + p.block_size) * (uint32_t)sizeof(TOut)), 0x00020000);
        float4 st = state[(size_t)grp * kWave + lane];
        if constexpr (DEPTH == 1) issue(slot0_t{}, rs_in, 0);
        consume(slot0_t{}); request(slot0_t{}, DEPTH * CS);
        for (uint32_t blk = 0; blk < nblk; ++blk) {
            const uint32_t n0 = blk * BLK;
            if constexpr (DEPTH == 2) {
#pragma unroll 1
                for (int q = 0; q < NCHUNK; q += 2) {
                    if (q != 0) { consume(slot0_t{}); request(slot0_t{}, n0 + CS * q + 2 * CS); }
                    busy();
                    consume(slot1_t{}); request(slot1_t{}, n0 + CS * (q + 1) + 2 * CS);
                    busy();
                }
            } else {
#pragma unroll 1
                for (int q = 0; q < NCHUNK; ++q) {
                    if (q != 0) { consume(slot0_t{}); request(slot0_t{}, n0 + CS * q + CS); }
                    busy();
                }
            }
            if (blk + 1 < nblk) { consume(slot0_t{}); request(slot0_t{}, n0 + BLK + DEPTH * CS); }
#pragma unroll 4
            for (int r = 0; r < CH; ++r) {
#pragma unroll
                for (int h = 0; h < (BLK / 4 + 63) / 64; ++h) {
                    const int t = 4 * (lane + 64 * h);
                    const int voff = t < BLK ? t * (int)sizeof(TOut) : 0x40000000;
                    const int soff = (int)((r * p.out_stride + n0) * (uint32_t)sizeof(TOut));
                    if constexpr (sizeof(TOut) == 4) __builtin_amdgcn_raw_buffer_store_b128(u4v_cw{ acc, acc, acc, acc }, rs_out, voff, soff, 2);
                    else __builtin_amdgcn_raw_buffer_store_b64(u2v_cw{ acc, acc }, rs_out, voff, soff, 2);
                }
            }
        }
        st.x += __uint_as_float(acc & 1u);
        state[(size_t)grp * kWave + lane] = st;
        if (grp + gridDim.x >= ngroups) break;
        grp += gridDim.x; rs_in = rs_next; rs_next = in_rsrc(grp + gridDim.x);
    }
    if (acc == 0x12345678u) roof_lds[lane] = __uint_as_float(acc);
}

template <int NS, typename TIn, typename TOut>
static hipError_t cw_roof_launch(const RxParams &p, const void *src, void *dst, float4 *state, uint32_t work, hipStream_t st)
{
    constexpr int CH = CwGeo<NS>::CH;
    const uint32_t ngroups = (p.channels + CH - 1) / CH;
#define CW_ROOF_BLK(B_)                                                                                                                        \
    if (p.block == B_) {                                                                                                                       \
        /* the residency of the kernel it stands for: as much dynamic LDS as leaves that many workgroups on a CU */                          \
        static size_t lds = 0;                                                                                                                 \
        static int resident = 0;                                                                                                               \
        if (lds == 0) {                                                                                                                        \
            int per_cu = 0;                                                                                                                    \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cw_fused<NS, 2, B_, TIn, TOut>, 64, 0) != hipSuccess || per_cu < 1)    \
                per_cu = 8;                                                                                                                    \
            lds = ((size_t)(160 * 1024) / (size_t)per_cu) & ~(size_t)1023;                                                                     \
            resident = cw_resident(k_cw_roof<NS, B_, TIn, TOut>, lds);                                                                         \
        }                                                                                                                                      \
        hipLaunchKernelGGL((k_cw_roof<NS, B_, TIn, TOut>), dim3(cw_grid(ngroups, resident)), dim3(64), lds, st, p,                              \
                           static_cast<const TIn *>(src), static_cast<TOut *>(dst), state, work);                                              \
        return hipGetLastError();                                                                                                              \
    }
    CW_ROOF_BLK(256)
    CW_ROOF_BLK(128)
    if constexpr (NS != 2) { CW_ROOF_BLK(512) }
#undef CW_ROOF_BLK
    return hipErrorNotSupported;
}

// state: channels rounded up to whole workgroups x NS float4 (scratch of the caller's)
hipError_t launch_cw_roof(const RxParams &p, const void *src, bool q15, void *dst, float4 *state, uint32_t work, hipStream_t st)
{
    if (!cw_strides_ok(p.in_stride, p.out_stride)) return hipErrorNotSupported;
    if (p.nbiq == 4) return q15 ? cw_roof_launch<4, int16_t, int16_t>(p, src, dst, state, work, st) : cw_roof_launch<4, float, float>(p, src, dst, state, work, st);
    if (p.nbiq == 2) return q15 ? cw_roof_launch<2, int16_t, int16_t>(p, src, dst, state, work, st) : cw_roof_launch<2, float, float>(p, src, dst, state, work, st);
    if (p.nbiq == 8) return q15 ? cw_roof_launch<8, int16_t, int16_t>(p, src, dst, state, work, st) : cw_roof_launch<8, float, float>(p, src, dst, state, work, st);
    return hipErrorNotSupported;
}

}  // namespace srx
