// rx_split16_kernels.h -- the SELENITE_ARITH_SPLIT16 kernels (k_ssb_split16, k_hilb_split16) and their launchers, shared by
// rx_split16.hip (f32 slots, the no-decimator kernels, dispatch) and rx_split16_q15.hip (int16 slots).
#pragma once
#include "rx_fused_kernels.h"

#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace srx {

// streaming state of a channel: written by one call and read back by the next -- 0.17 GB for 65 536 channels, which the 256 MB
// Infinity Cache can hold between calls if nothing streams through it: plain (cacheable) accesses (non-temporal ones measured
// 1.6 % slower: they defeat exactly that residency)
__device__ __forceinline__ float st_ld(const float *p)
{
    return *p;
}
__device__ __forceinline__ void st_st(float *p, float v)
{
    *p = v;
}

// sticky per-channel counters do not wrap (advisor, round 3: a uint32 of guarded blocks wraps after days at firmware slot rates)
__device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) { return a > 0xFFFFFFFFu - b ? 0xFFFFFFFFu : a + b; }

__device__ __forceinline__ float amax2(v2f x, float m) { return fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), m); }

// memory-order point for the single-wave workgroups of this file: LDS operations of a wave execute in
// issue order, so all that is needed is that the compiler keeps the program order of the memory
// operations on either side.  Unlike wave_lds_sync() this is NOT a scheduling barrier for ALU and
// matrix instructions: the phases on either side may overlap.
__device__ __forceinline__ void lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// AGC of one pass (256 audio samples, 4 per lane): arm_abs + arm_max per DSP block of GROUP lanes,
// gain law, arm_scale with the updated gain.  GROUP = 16 / 64: lane reductions by DPP, the block
// envelopes broadcast by v_readlane; GROUP = 0: any power-of-two `group` (run time).
// Parity guard: `gd.thr` = guard ratio x the largest |component| the pass's matrix product saw; every DSP block of the pass
// whose envelope (max |audio| before the gain) is below it is counted in gd.n (gd.first: the first lane of every DSP block).
// gd.hist / gd.nh (k_ssb_split16, SELENITE_ARITH_AUTO): the guarded blocks among the first ones of a call, those that still see the
// Hilbert-pair history the previous call left -- a history of split16 precision when that call kept the channel on the matrix
// kernel: the "handover" blocks of DESIGN.md section 3, which the exact rerun cannot make exact; they are counted on their own.
struct GuardPass {
    float thr;
    uint64_t first;
    uint32_t n;
    uint64_t hist;     // lanes of the blocks inside the Hilbert history of the call's start while pass 0 is demodulated, else 0
    uint32_t nh;
    // round 4 (advisor finding): the first blocks of a pass -- those inside the reach of the Hilbert-pair history, lanes `hm` -- also read
    // decimated samples the PASS BEFORE produced, whose split-precision error scales with the largest sample THAT pass's product saw:
    // their threshold thr_h = ratio x max(this pass's maximum, the previous pass's) (at a call's start: the level the call before left
    // in the channel's word).  A loud signal that ends up to nd + M (nh - 1) samples in front of a pass is then still guarded.
    float thr_h;
    uint64_t hm;
};
template <int GROUP>
__device__ __forceinline__ void agc_pass(const AgcParams &ap, uint32_t agc_on, int lane, int group, float (&au)[4], float &gain,
                                         int nvb, GuardPass &gd,   // nvb: DSP blocks of the pass that exist (a call's last pass may be partial)
                                         float m_lane = -1.0f)     // >= 0: max |.| of FOUR samples of this lane's 16-lane row of the pass, taken by the caller (GROUP 16 / 32 / 64)
{
    // (k_hilb_split16 takes the maxima in the matrix layout, where lane l holds samples 64 (l>>4) + 16 r + (l&15): the same 16-lane rows as the
    // store layout's 4 l + r, so every row / half-wave / wave maximum is the same number -- and the gain law no longer waits for the LDS transpose)
    float m = m_lane >= 0.0f ? m_lane : fmaxf(fmaxf(fabsf(au[0]), fabsf(au[1])), fmaxf(fabsf(au[2]), fabsf(au[3])));
    float g = gain, mine = gain;
    auto guard = [&](float env) {                    // env: the block envelope, in (at least) the first lane of every block
        const int lanes = nvb * (GROUP ? GROUP : group);
        const uint64_t exist = lanes >= 64 ? ~0ull : ((1ull << lanes) - 1ull);
        const uint64_t hit = (__builtin_amdgcn_ballot_w64(env < gd.thr) | (__builtin_amdgcn_ballot_w64(env < gd.thr_h) & gd.hm)) & gd.first & exist;
        gd.n += (uint32_t)__builtin_popcountll(hit);
        gd.nh += (uint32_t)__builtin_popcountll(hit & gd.hist);
    };
    if constexpr (GROUP == 16) {
        m = row16_fmax(m);
        guard(m);
        const float d = agc_desired(ap, m);          // one division sequence serves the four blocks
        float ds[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) ds[b] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), 16 * b));
        const int myblk = lane >> 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float gn = agc_step(ap, g, ds[b]);
            g = b < nvb ? gn : g;                    // blocks past the end of the call leave the gain alone
            mine = (b == myblk) ? g : mine;
        }
    } else if constexpr (GROUP == 32) {
        // two DSP blocks of two 16-lane rows each (decimation by 2): row maxima by DPP, the two rows of a block joined on
        // the scalar unit (|.| >= 0: the bit patterns order like the values), one division sequence for both blocks
        m = row16_fmax(m);
        const uint32_t r0 = __builtin_amdgcn_readlane(__float_as_uint(m), 0), r1 = __builtin_amdgcn_readlane(__float_as_uint(m), 16);
        const uint32_t r2 = __builtin_amdgcn_readlane(__float_as_uint(m), 32), r3 = __builtin_amdgcn_readlane(__float_as_uint(m), 48);
        const float e0 = __uint_as_float(r0 > r1 ? r0 : r1), e1 = __uint_as_float(r2 > r3 ? r2 : r3);
        guard(lane < 32 ? e0 : e1);
        const float d = agc_desired(ap, lane < 32 ? e0 : e1);
        const float ds[2] = { __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), 0)),
                              __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), 32)) };
        const int myblk = lane >> 5;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float gn = agc_step(ap, g, ds[b]);
            g = b < nvb ? gn : g;
            mine = (b == myblk) ? g : mine;
        }
    } else if constexpr (GROUP == 64) {
        m = __uint_as_float(wave_umax_bits_dpp(m));
        guard(m);
        g = agc_step(ap, g, agc_desired(ap, m));
        mine = g;
    } else {
        if ((group & 15) == 0) {
            // whole 16-lane rows per DSP block (blocks of 64 / 128 / 192 / 256 audio samples at run time: BASELINE cfg2 in DSP blocks of 192 frames is
            // one block of three rows per pass): row maxima by DPP, the rows of a block joined on the scalar unit (|.| >= 0: the bit patterns
            // order like the values) -- instead of up to six LDS round trips (ds_bpermute) of the general scan below
            m = row16_fmax(m);
            const uint32_t r0 = __builtin_amdgcn_readlane(__float_as_uint(m), 0), r1 = __builtin_amdgcn_readlane(__float_as_uint(m), 16);
            const uint32_t r2 = __builtin_amdgcn_readlane(__float_as_uint(m), 32), r3 = __builtin_amdgcn_readlane(__float_as_uint(m), 48);
            const int rows = group >> 4;                     // rows per block; row i belongs to block i / rows (a last, partial block is never looked at)
            uint32_t e0 = r0, e1 = r1, e2 = r2, e3 = r3;
            if (rows == 2) { e0 = e1 = max(r0, r1); e2 = e3 = max(r2, r3); }
            else if (rows == 3) { e0 = e1 = e2 = max(max(r0, r1), r2); }
            else if (rows >= 4) { e0 = e1 = e2 = e3 = max(max(r0, r1), max(r2, r3)); }
            const int row = lane >> 4;
            m = __uint_as_float(row == 0 ? e0 : (row == 1 ? e1 : (row == 2 ? e2 : e3)));
        } else if (group == 8) {
            // eight lanes per DSP block (decimation by 8 of 256-frame blocks, round 4): two quad permutes and a half-row mirror on the
            // DPP path instead of three LDS round trips (ds_bpermute)
            m = fmaxf(m, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
            m = fmaxf(m, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
            m = fmaxf(m, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0x141, 0xf, 0xf, false)));   // row_half_mirror
        } else if ((group & (group - 1)) == 0) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
                if (off < group) m = fmaxf(m, __shfl_xor(m, off, 64));
        } else {                                     // e.g. 6 lanes: the firmware's 96-frame blocks by 4 (dsp_if.h:69-73)
            // downward segmented scan: after ceil(log2(group)) steps the FIRST lane of every block holds the block maximum
            const int seg_end = (lane / group + 1) * group;
            for (int off = 1; off < group; off <<= 1) {
                const float v = __shfl_down(m, off, 64);
                m = lane + off < seg_end ? fmaxf(m, v) : m;
            }
        }
        guard(m);
        const float d = agc_desired(ap, m);          // one division sequence serves every block of the pass
        const int nblk = min(64 / group, nvb), myblk = lane / group;
        for (int b = 0; b < nblk; ++b) {             // (b * group is wave-uniform: v_readlane with a scalar lane select)
            g = agc_step(ap, g, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), b * group)));
            if (b == myblk) mine = g;
        }
    }
    gain = agc_on ? g : gain;
    mine = agc_on ? mine : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) au[r] = au[r] * mine;
}

// ------------------------------------------------------------------------------------------
// k_ssb_split16
//
// One wavefront per channel, passes of 1024 complex inputs -> 256 audio samples, software-pipelined:
//
//   mix(p):    buffer-loaded I/Q (prefetched a pass ahead) x LO (shared table, prefetched) by packed
//              complex multiplies; wave maximum -> block exponent; x 2^s; f16 hi/lo split straight
//              into the four LDS images (I/Q x hi/lo, 64-sample rows padded to 80 halfs: the A-fragment
//              ds_read_b128 of lane l sits at 160 (l&15) + 16 (l>>4) + imm, conflict free); the last
//              ND-1 mixed samples also as f32 (history copy).
//   mfma(p):   KS k-steps x 6 MFMAs (big: hi*hi; small: hi*lo + lo*hi; both rails), B fragments resident
//              in 8*KS VGPRs, A fragments read one k-step ahead.
//   demod(p-1) runs in the SAME basic block as mfma(p): Hilbert FIR on Q (structural zeros skipped), unit
//              delay on I, sideband combine, AGC (DPP / readlane), audio store.  Nothing but true data
//              dependencies orders the two, so the VALU work of one pass fills the issue slots under
//              the matrix work of the next.
//   The pass loop body is branch-free apart from the (rare) history re-split: no exec-masked copy
//   loops, no conditional prefetch (buffer range check instead), state written after the loop.
// ------------------------------------------------------------------------------------------
template <int NCO, int ND, int M, int NH, typename TIn, typename TOut, int AM, int GROUP, int FLAVOUR = 0>
__global__ __launch_bounds__(64, 2) void k_ssb_split16(RxParams p, FusedArgs fa, const TIn *__restrict__ src,
                                                       TOut *__restrict__ dst)
{
    // FLAVOUR: 0 the plain kernel; 1 (GROUP == 16) phase 1 of the global-gain call, which also leaves the block maxima behind; 2 (GROUP == 0)
    // decimation by 2 M on the by-M product (FusedArgs::dec2) -- instantiations of their own: compiled into the plain run-time-geometry
    // kernel, the by-8 bookkeeping cost the firmware's one-slot calls 13 % (a kernel already short of scalar registers)
    constexpr int ENV = FLAVOUR == 1 ? 1 : 0;
    constexpr bool DEC2 = FLAVOUR == 2;
    static_assert(!DEC2 || GROUP == 0, "decimation by 2 M: run-time geometry only");
    using G = Geo<ND, M, NH>;
    using GS = GeoS<NCO, ND, M, NH>;
    using R = BRaw<TIn>;
    using W = BOut<TOut>;
    static_assert(ND > 0 && (M == 4 || M == 2) && NH > 0 && G::T % 128 == 0 && GS::HS % 128 == 0, "split16 decimator: /4 or /2, + Hilbert");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    // FM (round 4; SELENITE_ARITH_AUTO only): a run-time flavour of the AM instantiations, as in k_ssb_fused -- the discriminator
    // z[n] conj(z[n-1]) on the decimated rails, z[-1] of a pass = the newest entry of the FIR pair's delay lines, which FM keeps running
    // (history moves, state write-back and the samples for k_hist_exact as in the SSB modes).  `keeps` = the FIR-pair history moves.
    const bool fm = AM != 0 && fa.am == 2u;                       // wave-uniform
    // (round 4) the instance's decimator may be shorter than the kernel's: p.nd <= ND taps, zero-padded in front (rx_fused.hip:
    // split16_template_nd); the state holds p.nd - 1 samples per rail, Fr = HQ4 M + 1 - p.nd slots into the kernel's history
    const bool keeps = AM == 0 || fm;
    // persistent: this workgroup runs channels blockIdx.x, blockIdx.x + gridDim.x, ... -- in SELENITE_ARITH_AUTO minus the channels the
    // exact kernel HOLDS (kFlagHold, round 4: a channel that was recomputed stays with the exact kernel until a call of it shows no
    // block near the guard ratio; the matrix kernel does not touch it -- no loads, no passes, its word stays).  The words of the next
    // 64 candidates are read with one wave load and kept as a scalar bit mask (bit k: channel ch_lo + k gridDim.x is ours).
    // (ch_lo: channel of the mask's bit 0; the next scan starts 64 candidates on -- saturating at the channel count, which fits 32 bits)
    uint64_t ch_mask = 0ull;
    uint32_t ch_lo = blockIdx.x;
    auto ch_scan = [&](uint32_t &w, bool &mine) {                 // issue: the words of the 64 candidates from ch_lo on
        const uint64_t ci = (uint64_t)ch_lo + (uint64_t)lane * gridDim.x;
        mine = ci < (uint64_t)p.channels;
        w = p.rerun_flag != nullptr ? p.rerun_flag[mine ? ci : 0] : 0u;
    };
    auto ch_take = [&](uint32_t w, bool mine) {                   // consume
        ch_mask = __builtin_amdgcn_ballot_w64(mine && (w & kFlagHold) == 0u);
    };
    auto ch_next = [&]() -> uint32_t {                            // the next channel of this workgroup, or p.channels when there is none
        while (ch_mask == 0ull) {                                 // wave-uniform; the first mask is taken below, a reload happens after 64 candidates
            const uint64_t nx = (uint64_t)ch_lo + 64ull * gridDim.x;
            if (nx >= (uint64_t)p.channels) return p.channels;
            ch_lo = (uint32_t)nx;
            uint32_t w; bool mine;
            ch_scan(w, mine);
            ch_take(w, mine);
        }
        const uint32_t k = (uint32_t)__builtin_ctzll(ch_mask);
        ch_mask &= ch_mask - 1ull;
        return ch_lo + k * gridDim.x;
    };
    uint32_t w_first; bool mine_first;
    ch_scan(w_first, mine_first);                                 // (in flight under the fragment loads below)
    // (the first channel of the sequence is blockIdx.x unless the exact kernel holds it: its first pass is prefetched at once, as in round 3,
    // and prefetched again for the right channel in that rare case -- waiting for the words first cost 1 % of the headline)
    uint32_t c = blockIdx.x, c_nx = 0u;
    float *tab = lds + GS::oTab;
    _Float16 *X = reinterpret_cast<_Float16 *>(lds + GS::oX);     // [rail][hi/lo][IMG]
    v2f *Hf = reinterpret_cast<v2f *>(lds + GS::oHf);             // f32 (I, Q) history, HS samples
    float *D = lds + GS::oD;
    float *dI = D, *dQ = D + G::DLEN;
    constexpr int NLD = G::T / 128;                               // loads per lane per pass
    constexpr int NTL = GS::HS / 128;                             // ... of which the last NTL hold the next history
    static_assert(GS::HS % 128 == 0 && NTL >= 1 && NTL <= NLD, "history is a whole number of wave loads");

    // passes of the call; the last one may be partial (a call is a whole number of DSP blocks, not of passes): its
    // missing input reads as zeros and its surplus audio is dropped by the buffer range checks, the streaming state
    // and the AGC are taken from the part that exists (the host side sends such calls here only when that part holds
    // a whole decimator history: tail_in >= HS)
    // a full pass produces pq audio samples: 256, or (GROUP == 0 launches only) the largest whole number of DSP blocks in 256 when the
    // block does not divide it -- 240 for the firmware's 96-frame blocks by 4 (dsp_if.h:69-73).  Such passes run the "partial" code
    // path every time: the tile is computed in full, 16 of its outputs are dropped, the histories advance by pq * M samples.
    // DECIMATION BY 2 M (round 4, late: by 8 on the by-4 kernel).  A by-8 Toeplitz band is nd + 15 x 8 wide -- twelve k-steps, 96 VGPRs of
    // fragments -- and a 256-output pass would be 2048 inputs, sixteen loads in flight: neither fits this one-wave pipeline.  But the
    // by-8 outputs are every second by-4 output: the SAME 1024-input tile, images, fragments and k-steps as the by-4 kernel, with half of
    // the tile's 256 results dropped where the accumulators are written to LDS (`dwrite`) -- the matrix work per INPUT sample is that of
    // the by-4 chain, the demodulator's halves.  Everything behind the decimator runs on pass_out <= 128 outputs per pass through the
    // run-time geometry (GROUP == 0) that the 240- / 192-output passes already use.  fa.dec2: 0 off, 1 / 2 the even / odd outputs.
    const uint32_t dec2 = DEC2 ? fa.dec2 : 0u;                      // wave-uniform; a compile-time 0 in every other instantiation
    const uint32_t MO = dec2 ? 2u * (uint32_t)M : (uint32_t)M;      // input samples per audio sample
    const uint32_t pq = GROUP == 0 ? fa.pass_out : (uint32_t)G::P, tq = pq * MO;
    const uint32_t npass = (p.nout + pq - 1) / pq;
    const uint32_t tail_out = p.nout - (npass - 1) * pq;         // audio samples of the last pass: pq when the call is whole passes
    auto cur_out = [&](uint32_t pass) { return pass + 1 == npass ? tail_out : pq; };
    auto in_rsrc = [&](uint32_t ch) {                             // a channel past the last one: empty range, loads return zeros
        return make_rsrc(src + (size_t)ch * p.in_stride * 2, ch < p.channels ? p.block_size * (R::kBytes / 2) : 0u);
    };
    __amdgpu_buffer_rsrc_t rs_in = in_rsrc(c), rs_in_next = in_rsrc(c);      // (rs_in_next: set once the channel sequence is known)
    __amdgpu_buffer_rsrc_t rs_out = make_rsrc(dst + (size_t)c * p.out_stride, p.nout * (W::kBytes / 4));
    // global gain, phase 1: the kernel runs with its own AGC off and leaves max |audio| of every DSP block of every channel
    // behind, so the envelope reduction does not have to read the audio again (GROUP == 16 launches with whole passes only)
    // (ENV: its own instantiation -- the extra descriptor and branch cost the plain kernel 1.3 % when they were always compiled in)
    const uint32_t env_nblk = ENV ? p.block_size / p.block : 0u;
    auto env_rsrc = [&](uint32_t ch) { return make_rsrc(p.env_part + (size_t)ch * env_nblk, env_nblk * 4u); };
    __amdgpu_buffer_rsrc_t rs_env = env_rsrc(ENV ? c : 0u);
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(p.lo, NCO == 2 ? p.block_size * 8u : 0u);
    const int kInPass = (int)tq * (R::kBytes / 2);                // input bytes a pass advances by

    typename R::type raw[NLD];
    // shared LO (NCO == 2): an L2-resident table, so only LOD wave loads are kept in flight: the first LOD of a
    // pass are issued with the pass's input prefetch, the others by the mix stage as it frees the slots
    constexpr int LOD = 3;
    u4v lo4[LOD];
    auto lo_load = [&](int slot, int i, int sl) {
        lo4[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, lane * 16 + i * 1024, sl, 0);
    };
    auto lo_base = [&](uint32_t pass) { return pass < npass ? (int)(pass * tq) * 8 : 0; };   // pass == npass: the next channel's pass 0
    // periodic shared LO (NCO == 3: the table repeats every 256 samples and a pass is a whole number of periods): load
    // i of any pass multiplies by LO[(128 i + 2 lane, + 1) mod 256] -- two register quads for the whole kernel
    // (NCO == 4: the same for a PER-CHANNEL step that is a multiple of 2^24 -- every channel on the fs / 256 grid with its own LO:
    // the two quads are computed once per channel, in install_state, with the arithmetic of the per-sample NCO (nco_lo_pair))
    u4v lo_per[2];
    static_assert((NCO != 3 && NCO != 4) || G::T % 256 == 0, "a pass is a whole number of LO periods");
    if constexpr (NCO == 3) {
        lo_per[0] = *reinterpret_cast<const u4v *>(p.lo + 2 * lane);
        lo_per[1] = *reinterpret_cast<const u4v *>(p.lo + 128 + 2 * lane);
    }
    auto prefetch = [&](uint32_t pass) {                          // pass == npass: pass 0 of this workgroup's next channel
        const int so = pass < npass ? (int)pass * kInPass : 0;
        const __amdgpu_buffer_rsrc_t rs = pass < npass ? rs_in : rs_in_next;
#pragma unroll
        for (int i = 0; i < NLD; ++i) raw[i] = R::load(rs, lane * R::kBytes + i * 64 * R::kBytes, so);      // (BRaw: aux = SRX_IN_AUX)
        if constexpr (NCO == 2) {
#pragma unroll
            for (int i = 0; i < LOD; ++i) lo_load(i, i, lo_base(pass));
        }
    };
    prefetch(0);
    // Toeplitz B fragments (8 halfs per lane): [kk][hi/lo]
    h8 Bh[GS::KS], Bl[GS::KS];
    {
        const h8 *bt = static_cast<const h8 *>(fa.btab16);
#pragma unroll
        for (int kk = 0; kk < GS::KS; ++kk) {
            Bh[kk] = bt[(2 * kk + 0) * 64 + lane];
            Bl[kk] = bt[(2 * kk + 1) * 64 + lane];
        }
    }
    // Hilbert taps: wave-uniform values held in scalar registers for the whole kernel (only the structurally
    // non-zero ones are ever referenced: (NH + 1) / 2 SGPRs), so a tap costs no v_readlane per pass
    float hreg[(NH + 63) / 64];
#pragma unroll
    for (int v = 0; v < (NH + 63) / 64; ++v) hreg[v] = (64 * v + lane < NH) ? p.hilb_c[64 * v + lane] : 0.0f;
    ch_take(w_first, mine_first);
    {
        const uint32_t c_real = ch_next();
        if (c_real >= p.channels) return;                         // (every channel of this workgroup is held)
        if (c_real != c) {                                        // wave-uniform, rare
            c = c_real;
            rs_in = in_rsrc(c);
            rs_out = make_rsrc(dst + (size_t)c * p.out_stride, p.nout * (W::kBytes / 4));
            if constexpr (ENV != 0) rs_env = env_rsrc(c);
            prefetch(0);
        }
        c_nx = ch_next();
    }
    rs_in_next = in_rsrc(c_nx);

    auto htap = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hreg[k >> 6]), k & 63)); };
    if constexpr (NCO == 1 || NCO == 4)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];

    // ---- streaming state of a channel: loaded into registers (for the next channel of this workgroup while
    // the current one is still in its last pass), installed into LDS when the channel starts.  Flat history
    // sample f in [0, HS) is CMSIS state sample s = f - F (older slots meet zero taps only); all loads are
    // unconditional from a clamped index (one memory round trip for the lot).
    constexpr int NHI = GS::HS / kWave, NFI = 2 * G::HH4 / kWave;
    static_assert(GS::HS % kWave == 0 && (2 * G::HH4) % kWave == 0, "state fills are whole wave loads");
    v2f st_hv[NHI];
    float st_fv[NFI], st_gain;
    uint32_t st_ph0, st_step, st_word;
    auto load_state = [&](uint32_t ch) {
        const int ndr = (int)p.nd, Fr = G::HQ4 * M + 1 - ndr;
        ch = ch < p.channels ? ch : p.channels - 1;               // past the last channel: harmless reload, never installed
        const float *stI = p.dec_state + (size_t)ch * 2 * (ndr - 1), *stQ = stI + (ndr - 1);
        const float *stF = p.fir_state + (size_t)ch * 2 * G::HH;
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int s = j * kWave + lane - Fr, sc = s < 0 ? 0 : s;
            const float xi = st_ld(stI + sc), xq = st_ld(stQ + sc);
            st_hv[j] = s < 0 ? v2f{ 0.0f, 0.0f } : v2f{ xi, xq };
        }
#pragma unroll
        for (int j = 0; j < NFI; ++j) {
            const int i = j * kWave + lane;
            const int rail = i / G::HH4, sidx = i % G::HH4 - G::FH;
            const float x = st_ld(stF + rail * G::HH + (sidx < 0 ? 0 : sidx));
            st_fv[j] = sidx < 0 ? 0.0f : x;
        }
        st_ph0 = NCO ? p.phase[ch] : 0u;
        st_step = NCO ? p.step[ch] : 0u;
        st_gain = p.gain[ch];
        st_word = p.rerun_flag ? p.rerun_flag[ch] : 0u;           // AUTO: provenance of the channel's state, which hist_ext buffer goes with it
    };
    uint32_t prev_prov = kProvExact, prev_buf = 0u, st_word_cur = 0u;
    // SELENITE_ARITH_AUTO: the mixed samples in front of the decimator state go to hist_ext (RxParams), so that a rerun of the NEXT
    // call can start from a Hilbert-pair history in exact arithmetic (k_hist_exact) -- when this call is long enough to hold them
    // (the row holds positions [E - (ND - 1) - ext_len + 1, E - (ND - 1) + 1): moved up by one sample so that it starts on an even
    // one -- the two samples of a lane leave in ONE 16-byte store that never straddles the window, a row is ext_len * 8 bytes of
    // whole cache lines -- which costs nothing: the oldest M (HH4 - HH) >= 2 samples of the nominal window meet no tap)
    static_assert(ND % 2 == 0 && M * (G::HH4 - G::HH) >= 1, "hist_ext rows are pair-aligned for even tap counts");
    const int ext_start = (int)p.block_size - ((int)p.nd - 1) - (int)p.ext_len + 1; // call-relative position of hist_ext[0] (even: p.nd is)
    // (AM never touches the Hilbert-pair history: it keeps no samples either, and what it hands on is the provenance it found --
    // degraded to "matrix kernel, no samples" when it was "with samples", because the decimator state moves on without them)
    const bool ext_on = keeps && p.hist_ext != nullptr && ext_start >= 0;     // wave-uniform
    __amdgpu_buffer_rsrc_t rs_ext = make_rsrc(p.hist_ext, 0u);
    uint32_t b_hist = 0, ph0 = 0, step = 0;                       // b_hist: bit pattern of the largest |history component|
    float gain = 1.0f;
    int s_cur = 0x7fff;                                               // sample scale exponent of the images (none yet)
    // parity guard (GuardPass): thresholds of the pass being mixed and of the pass before it (whose demodulator runs later)
    // thresholds (guard ratio x pass maximum) of the pass being mixed, of the one before it (whose demodulator runs under this pass's
    // matrix stage) and of the one before that (its first blocks read Hilbert-pair history from there); at a call's start thr_prev
    // comes from the level in the channel's word.  b_last: bit pattern of the last pass's maximum, for that word.
    float thr_cur = 0.0f, thr_prev = 0.0f, thr_pp = 0.0f;
    uint32_t b_last = 0u;
    GuardPass gd;
    gd.thr = 0.0f; gd.n = 0u; gd.hist = 0ull; gd.nh = 0u; gd.thr_h = 0.0f;
    if constexpr (GROUP == 16) gd.first = 0x0001000100010001ull;
    else if constexpr (GROUP == 32) gd.first = 0x0000000100000001ull;
    else if constexpr (GROUP == 64) gd.first = 1ull;
    else {
        gd.first = 0ull;
        for (int l = 0; l < 64; l += (int)fa.group) gd.first |= 1ull << l;
    }
    // lanes of the DSP blocks that hold one of the first HH audio samples of a call (the reach of the Hilbert-pair history)
    const int hist_lanes = ((G::HH + 4 * (int)fa.group - 1) / (4 * (int)fa.group)) * (int)fa.group;
    const uint64_t hist_mask = hist_lanes >= 64 ? ~0ull : ((1ull << hist_lanes) - 1ull);
    // (AM reads no history at all; FM one sample of it, z[-1]: the first DSP block of a pass)
    gd.hm = AM == 0 ? hist_mask : (fm ? ((GROUP ? GROUP : (int)fa.group) >= 64 ? ~0ull : ((1ull << (GROUP ? GROUP : (int)fa.group)) - 1ull)) : 0ull);
    if constexpr (GROUP != 0 && AM == 0) {                            // a compile-time constant where the DSP-block geometry is one (no scalar registers)
        constexpr int hl = ((G::HH + 4 * GROUP - 1) / (4 * GROUP)) * GROUP;
        gd.hm = hl >= 64 ? ~0ull : ((1ull << hl) - 1ull);
    }
    auto install_state = [&]() {
        float mh = 0.0f;
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            Hf[j * kWave + lane] = st_hv[j];
            mh = amax2(st_hv[j], mh);
        }
#pragma unroll
        for (int j = 0; j < NFI; ++j) {
            const int i = j * kWave + lane;
            D[(i / G::HH4) * G::DLEN + i % G::HH4] = st_fv[j];
        }
        b_hist = wave_umax_bits(mh);
        ph0 = st_ph0; step = st_step; gain = st_gain;
        s_cur = 0x7fff;
        gd.n = 0u; gd.nh = 0u;
        thr_cur = __uint_as_float(st_word & kLvlMask) * p.guard_ratio;     // ("the pass before pass 0": shifted into thr_prev by mix(0))
        thr_prev = 0.0f;
        prev_prov = (st_word >> kProvShift) & kProvMask; prev_buf = (st_word >> kExtBufShift) & 1u; st_word_cur = st_word;
        if (ext_on)                                                   // the buffer the state of the call before does NOT point at
            rs_ext = make_rsrc(p.hist_ext + (size_t)(prev_buf ^ 1u) * p.ext_buf_stride + (size_t)c * p.ext_len, p.ext_len * (sizeof(TIn) == 2 ? 4u : 8u));
        if constexpr (NCO == 4) {
            // LO of samples 2 lane, 2 lane + 1 and 128 + 2 lane, 129 + 2 lane of every 256-sample period: the phases the
            // per-sample NCO (NCO == 1) would form for them in any pass, n0 * step and 256 * step being multiples of 2^32
            const uint32_t pe = ph0 + 2u * lane * step;
            v2f la, lb;
            nco_lo_pair(tab, pe, pe + step, la, lb);
            lo_per[0] = u4v{ __float_as_uint(la.x), __float_as_uint(la.y), __float_as_uint(lb.x), __float_as_uint(lb.y) };
            nco_lo_pair(tab, pe + 128u * step, pe + 129u * step, la, lb);
            lo_per[1] = u4v{ __float_as_uint(la.x), __float_as_uint(la.y), __float_as_uint(lb.x), __float_as_uint(lb.y) };
        }
    };
    load_state(c);
    const int group = (int)fa.group;
    const int abase = GS::RSTR * (lane & 15) + 8 * (lane >> 4);       // A-fragment lane base (halfs): row l&15 of the output tile

    // two (I, Q) samples f (even), f + 1, times the block scale `pre` -> one word in each of the four images:
    //   hi = f16(x * pre), lo = f16(x * pre - hi), one v_fma_mixlo/hi_f16 each (the product with the power of
    // two and the difference are exact inside the fused operation, so this IS convert / subtract / convert,
    // bit for bit: tools/mix_split_check.hip) -- 8 plain vector instructions per sample pair instead of 12
    // (4 of them packed) for scale, convert, convert back, subtract, convert.
    auto put_iq = [&](int f, v2f a, v2f b, float pre) {
        uint32_t hI, hQ, lI, lQ;
        const v2f pre2 = { pre, pre };
        const v2f sa = a * pre2, sb = b * pre2;
        const h2 hhI = __builtin_convertvector(v2f{ sa.x, sb.x }, h2), hhQ = __builtin_convertvector(v2f{ sa.y, sb.y }, h2);
        const v2f ra = sa - v2f{ (float)hhI.x, (float)hhQ.x }, rb = sb - v2f{ (float)hhI.y, (float)hhQ.y };
        const h2 llI = __builtin_convertvector(v2f{ ra.x, rb.x }, h2), llQ = __builtin_convertvector(v2f{ ra.y, rb.y }, h2);
        hI = __builtin_bit_cast(uint32_t, hhI); hQ = __builtin_bit_cast(uint32_t, hhQ);
        lI = __builtin_bit_cast(uint32_t, llI); lQ = __builtin_bit_cast(uint32_t, llQ);
        const int ph = GS::phys(f);
        *reinterpret_cast<uint32_t *>(X + 0 * GS::IMG + ph) = hI;
        *reinterpret_cast<uint32_t *>(X + 1 * GS::IMG + ph) = lI;
        *reinterpret_cast<uint32_t *>(X + 2 * GS::IMG + ph) = hQ;
        *reinterpret_cast<uint32_t *>(X + 3 * GS::IMG + ph) = lQ;
    };

    // ---- mix(p): NCO mix, block exponent, f16 split into the images, f32 history copy ----
    auto mix = [&](uint32_t pass, auto partial_c) {
        constexpr bool PARTIAL = decltype(partial_c)::value;     // the call's last pass, with fewer than T input samples
        const uint32_t n0 = pass * tq;
        const uint32_t cur_in = cur_out(pass) * MO;                   // input samples of this pass that belong to it
        const uint32_t ph_lane = ph0 + 2u * lane * step;
        v2f m[2 * NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            v2f a, b;
            R::unpack(raw[i], a, b);
            if constexpr (NCO == 2) {
                const u4v l = lo4[i % LOD];
                cmul_pk2(a, b, v2f{ __uint_as_float(l.x), __uint_as_float(l.y) }, v2f{ __uint_as_float(l.z), __uint_as_float(l.w) },
                         m[2 * i], m[2 * i + 1]);
                if (i + LOD < NLD) lo_load(i % LOD, i + LOD, lo_base(pass));
            } else if constexpr (NCO == 3 || NCO == 4) {
                const u4v l = lo_per[i & 1];
                cmul_pk2(a, b, v2f{ __uint_as_float(l.x), __uint_as_float(l.y) }, v2f{ __uint_as_float(l.z), __uint_as_float(l.w) },
                         m[2 * i], m[2 * i + 1]);
            } else
            if constexpr (NCO == 1) {
                // phase of sample n0 + 128 i + 2 lane: a per-channel lane term plus a wave-uniform term
                const uint32_t pe = ph_lane + (n0 + 128u * i) * step;
                v2f la, lb;
                nco_lo_pair(tab, pe, pe + step, la, lb);
                cmul_pk2(a, b, la, lb, m[2 * i], m[2 * i + 1]);
            } else {
                m[2 * i] = a;
                m[2 * i + 1] = b;
            }
        }
        if (ext_on && (int)(n0 + G::T) > ext_start) {                 // wave-uniform: the last one or two passes of the call
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e0 = (int)n0 + 128 * i - ext_start;               // row index of the load's first sample (even)
                // (opaque to the optimizer: the passes that meet the window are the same for every channel, and the eight lane offsets below,
                // hoisted out of the channel loop, cost the kernel its last free registers -- 40 bytes of scratch in a kernel that had none)
                asm volatile("" : "+s"(e0));
                if (e0 + 128 > 0 && e0 < (int)p.ext_len) {        // wave-uniform: this load meets the window (pairs outside it: out of range, dropped)
                    if constexpr (sizeof(TIn) == 2) {                 // int16 slots: the raw samples as they came (half the bytes; the NCO phase of every one is known)
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, raw[i]), rs_ext, (e0 + 2 * lane) * 4, 0, SRX_OUT_AUX);
                    } else {
                        const u4v pr = { __float_as_uint(m[2 * i].x), __float_as_uint(m[2 * i].y), __float_as_uint(m[2 * i + 1].x), __float_as_uint(m[2 * i + 1].y) };
                        __builtin_amdgcn_raw_buffer_store_b128(pr, rs_ext, (e0 + 2 * lane) * 8, 0, SRX_OUT_AUX);      // (non-temporal: read back only by a rerun)
                    }
                }
            }
        }
        float mt = 0.0f, mh = 0.0f;                                   // |.| maxima: tail (next history), head
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float &mm = (i >= NLD - NTL) ? mt : mh;
            mm = amax2(m[2 * i], mm);
            mm = amax2(m[2 * i + 1], mm);
        }
        const uint32_t b_tail = wave_umax_bits(mt);
        const uint32_t b_need = max(max(wave_umax_bits(mh), b_tail), b_hist);       // the largest |component| the pass's images hold
        const uint32_t e_need = b_need >> 23;
        thr_pp = thr_prev; thr_prev = thr_cur;
        thr_cur = __uint_as_float(b_need) * p.guard_ratio;
        b_last = b_need;
        // largest |component| * 2^s in [2^14, 2^15):  s = 14 - (E - 127); 2^s must itself be a normal float
        int s_new = 141 - (int)e_need;
        s_new = s_new > 127 ? 127 : (s_new < -126 ? -126 : s_new);
        if (s_new != s_cur) {                                         // wave-uniform; always taken in the first pass
            const float pre = __uint_as_float((uint32_t)(s_new + 127) << 23);
#pragma unroll
            for (int j = 0; j < GS::HS / 128; ++j) {
                const int f = 2 * (j * kWave + lane);
                const float4 hq = lds_ld4f(reinterpret_cast<const float *>(Hf + f));
                put_iq(f, v2f{ hq.x, hq.y }, v2f{ hq.z, hq.w }, pre);
            }
            s_cur = s_new;
        }
        // (a partial pass: the next history is not the last loads of the tile -- the maximum of the whole tile is a safe bound)
        b_hist = PARTIAL ? max(wave_umax_bits(mh), b_tail) : b_tail;
        lds_order();                                                  // history reads above, history writes below
        if constexpr (PARTIAL) {
            // a pass shorter than the decimator history (one-pass calls only: the host cuts longer calls so that this is the
            // whole call, e.g. one 96-frame slot): the next history keeps the last HS - cur_in samples of the present one
            if (cur_in < (uint32_t)GS::HS) {                          // wave-uniform
                float4 keep[GS::HS / 128];
#pragma unroll
                for (int j = 0; j < GS::HS / 128; ++j) {
                    const int f = 2 * (j * kWave + lane) + (int)cur_in;
                    keep[j] = f < GS::HS ? lds_ld4f(reinterpret_cast<const float *>(Hf + f)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                lds_order();
#pragma unroll
                for (int j = 0; j < GS::HS / 128; ++j) {
                    const int f = 2 * (j * kWave + lane);
                    if (f + (int)cur_in < GS::HS) *reinterpret_cast<float4 *>(Hf + f) = keep[j];
                }
            }
        }
        const float pre = __uint_as_float((uint32_t)(s_cur + 127) << 23);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int n = 128 * i + 2 * lane;
            put_iq(GS::HS + n, m[2 * i], m[2 * i + 1], pre);
            if constexpr (PARTIAL) {                                  // the history is the last HS samples that exist
                const int hidx = n - ((int)cur_in - GS::HS);
                if (hidx >= 0 && hidx < GS::HS)
                    *reinterpret_cast<float4 *>(Hf + hidx) = make_float4(m[2 * i].x, m[2 * i].y, m[2 * i + 1].x, m[2 * i + 1].y);
            } else if (i >= NLD - NTL) {
                *reinterpret_cast<float4 *>(Hf + (n - (G::T - GS::HS))) = make_float4(m[2 * i].x, m[2 * i].y, m[2 * i + 1].x, m[2 * i + 1].y);
            }
        }
    };

    // ---- mfma(p): the decimator, 3 f16 MFMAs per k-step and rail (hi*hi | hi*lo + lo*hi) ----
    // one f32 accumulator per rail takes the big (hi*hi) and the two small (hi*lo, lo*hi) terms: the small terms are
    // rounded at the accumulator's ulp as they arrive (~20 extra roundings of 2^-24 relative, against a 1e-5 bar)
    // and the sum never has to be formed on the vector ALU
    v4f accI, accQ;
    auto mfma_phase = [&](auto &&side) {
        accI = accQ = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
        const _Float16 *xIh = X + 0 * GS::IMG + abase, *xIl = X + 1 * GS::IMG + abase;
        const _Float16 *xQh = X + 2 * GS::IMG + abase, *xQl = X + 3 * GS::IMG + abase;
        auto offA = [](int kk) { return GS::phys(32 * kk); };                // fragments never straddle a row (GeoS)
        h8 aIh = *reinterpret_cast<const h8 *>(xIh + offA(0)), aIl = *reinterpret_cast<const h8 *>(xIl + offA(0));
        h8 aQh = *reinterpret_cast<const h8 *>(xQh + offA(0)), aQl = *reinterpret_cast<const h8 *>(xQl + offA(0));
#pragma unroll
        for (int kk = 0; kk < GS::KS; ++kk) {
            h8 nIh = aIh, nIl = aIl, nQh = aQh, nQl = aQl;
            if (kk + 1 < GS::KS) {
                const int off = offA(kk + 1);
                nIh = *reinterpret_cast<const h8 *>(xIh + off); nIl = *reinterpret_cast<const h8 *>(xIl + off);
                nQh = *reinterpret_cast<const h8 *>(xQh + off); nQl = *reinterpret_cast<const h8 *>(xQl + off);
            }
            accI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIh, Bl[kk], accI, 0, 0, 0);     // small terms first
            accQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQh, Bl[kk], accQ, 0, 0, 0);
            accI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIl, Bh[kk], accI, 0, 0, 0);
            accQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQl, Bh[kk], accQ, 0, 0, 0);
            accI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIh, Bh[kk], accI, 0, 0, 0);
            accQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQh, Bh[kk], accQ, 0, 0, 0);
            aIh = nIh; aIl = nIl; aQh = nQh; aQl = nQl;
            side(kk);                                                 // vector work that runs under this k-step's MFMAs
        }
    };
    // decimated rails of the pass into D behind the Hilbert history (exact power-of-two rescale)
    auto dwrite = [&]() {
        const int o0 = G::HH4 + 64 * (lane >> 4) + (lane & 15);
        const int ex = -(s_cur + fa.split_sc);
        if (dec2) {                                                   // wave-uniform: tile output 64 (l >> 4) + 16 r + (l & 15) is chain output (that - parity) / 2
            const int oh = G::HH4 + 32 * (lane >> 4) + ((lane & 15) >> 1);
            if ((uint32_t)(lane & 1) + 1u == dec2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dI[oh + 8 * r] = __builtin_ldexpf(accI[r], ex);
                    dQ[oh + 8 * r] = __builtin_ldexpf(accQ[r], ex);
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dI[o0 + 16 * r] = __builtin_ldexpf(accI[r], ex);
            dQ[o0 + 16 * r] = __builtin_ldexpf(accQ[r], ex);
        }
    };
    // history of the four images: last HS samples of the pass back to the front, 16 bytes per move
    constexpr int CPI = GS::HS / 8, CPR = GS::RL / 8;                 // 16-byte chunks per image history / per physical row
    constexpr int NCB = 4 * CPI / kWave;                              // moves per lane
    static_assert((4 * CPI) % kWave == 0 && GS::HS % GS::RL == 0, "image history is a whole number of wave moves and of rows");
    auto cb_addr = [&](int k) {                                       // halfs; chunk i -> (image, row, 8-half column group)
        const int i = k * kWave + lane, img = i / CPI, rem = i % CPI;
        return img * GS::IMG + GS::RSTR * (rem / CPR) + 8 * (rem % CPR);
    };
    auto cb_read = [&](u4v (&cb)[NCB]) {
#pragma unroll
        for (int k = 0; k < NCB; ++k) cb[k] = *reinterpret_cast<const u4v *>(X + cb_addr(k) + GS::RSTR * (int)(tq / GS::RL));      // (whole rows: the host sends pq * M % RL == 0 only)
    };
    auto cb_write = [&](const u4v (&cb)[NCB]) {
#pragma unroll
        for (int k = 0; k < NCB; ++k) *reinterpret_cast<u4v *>(X + cb_addr(k)) = cb[k];
    };
    // Hilbert-pair history: last HH4 decimated samples of both rails to the front of D (every lane moves
    // one float4; the upper half of the wave repeats the lower half's moves when 2*HH4/4 = 32)
    constexpr int NDV = 2 * (G::HH4 / 4);
    static_assert(NDV == 16 || NDV == 32 || NDV == 64, "Hilbert history move is one float4 per lane (lanes beyond NDV repeat the first ones)");
    const int dt_off = ((lane % NDV) / (G::HH4 / 4)) * G::DLEN + 4 * (lane % (G::HH4 / 4));

    // ---- demod: Hilbert pair + sideband (or AM envelope), AGC of the pass whose decimated rails are in D.
    // Cut into KS pieces that sit, in program order, behind the MFMAs of the k-steps of the NEXT pass's matrix
    // stage: TPK Hilbert read-and-accumulate steps per k-step, the rest (sideband, AGC) in the last pieces.
    constexpr int NTS = HilbertSteps<ND, M, NH>::N;
    constexpr int KH = GS::KS > 4 ? GS::KS - 3 : 1;                   // k-steps that carry Hilbert steps
    constexpr int TPK = (NTS + KH - 1) / KH;
    float q2[4];
    auto demod_piece = [&](int kk, float (&au)[4], int nvb = 64) {
        if constexpr (AM != 0) {
            if (kk == 0) {
                const float4 vi = lds_ld4f(dI + G::HH4 + 4 * lane);
                const float4 vq = lds_ld4f(dQ + G::HH4 + 4 * lane);
                if (fm) {
                    const float pi0 = dI[G::HH4 + 4 * lane - 1], pq0 = dQ[G::HH4 + 4 * lane - 1];
                    au[0] = fm_disc(vi.x, vq.x, pi0, pq0);  au[1] = fm_disc(vi.y, vq.y, vi.x, vq.x);
                    au[2] = fm_disc(vi.z, vq.z, vi.y, vq.y); au[3] = fm_disc(vi.w, vq.w, vi.z, vq.z);
                    // parity guard of the discriminator: its error is |dz| / (pi |z|) per sample of the pair, |dz| ~ 1e-6 of the pass
                    // maximum (the split product), against a bar of 1e-5 of the block's max |audio|: a block is guarded when
                    // min|z| x max|audio| < ratio x pass maximum (min|z| over the pass and the sample in front of it): the thresholds of
                    // this pass divided by min|z| (a pass that touches zero guards everything)
                    float zz = fminf(fminf(vi.x * vi.x + vq.x * vq.x, vi.y * vi.y + vq.y * vq.y), fminf(vi.z * vi.z + vq.z * vq.z, vi.w * vi.w + vq.w * vq.w));
                    zz = fminf(zz, pi0 * pi0 + pq0 * pq0);
                    const float zmin = __builtin_sqrtf(__uint_as_float(~wave_umax_bits(__uint_as_float(~__float_as_uint(zz)))));
                    gd.thr = gd.thr / zmin;
                    gd.thr_h = gd.thr_h / zmin;
                } else {
                    au[0] = cmag<0>(vi.x, vq.x); au[1] = cmag<0>(vi.y, vq.y);
                    au[2] = cmag<0>(vi.z, vq.z); au[3] = cmag<0>(vi.w, vq.w);
                }
                agc_pass<GROUP>(p.agcp, p.agc, lane, group, au, gain, nvb, gd);
            }
        } else {
            if (kk == 0) q2[0] = q2[1] = q2[2] = q2[3] = 0.0f;
#pragma unroll
            for (int j = 0; j < TPK; ++j)
                if (kk * TPK + j < NTS) hilbert_tstep<1, ND, M, NH>(kk * TPK + j, dQ, lane, htap, q2);
            if (kk == (NTS + TPK - 1) / TPK - 1 || (kk == GS::KS - 1 && (NTS + TPK - 1) / TPK > GS::KS)) {
                const float *di = dI + G::FH + fa.delay_idx + 4 * lane;   // unit-impulse delay FIR
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float i2 = di[r] + 0.0f;                    // 0.0f + 1.0f*x of the dense loop
                    au[r] = fa.upper ? (i2 - q2[r]) : (i2 + q2[r]);   // arm_sub_f32 / arm_add_f32
                }
                agc_pass<GROUP>(p.agcp, p.agc, lane, group, au, gain, nvb, gd);
            }
        }
    };
    static_assert(TPK * GS::KS >= NTS, "every Hilbert step has a k-step");
    auto demod = [&](float (&au)[4], int nvb) {                       // the whole demodulator in one piece (last pass)
#pragma unroll
        for (int kk = 0; kk < GS::KS; ++kk) demod_piece(kk, au, nvb);
    };
    bool nonfinite = false;                                           // any audio sample of this workgroup NaN / Inf (x * 0 is NaN iff x is)
    auto store_audio = [&](uint32_t q, const float (&au)[4]) {
        const float z = __builtin_fmaf(au[3], 0.0f, __builtin_fmaf(au[2], 0.0f, __builtin_fmaf(au[1], 0.0f, au[0] * 0.0f)));
        nonfinite = nonfinite || ((z != z) && (!DEC2 || (uint32_t)lane < pq / 4u));            // (by 2 M: the lanes past the pass hold no output of the chain)
        // (pq < 256: the last lanes hold outputs of the NEXT pass's region, computed from its first samples -- not stored)
        const int vo = (GROUP != 0 || (uint32_t)lane < pq / 4u) ? lane * W::kBytes : 0x40000000;
        W::store(rs_out, vo, (int)(q * pq) * (W::kBytes / 4), au, ENV != 0, p.q15_round);      // (ENV: phase 1 of the global-gain call -- the gain pass reads this audio back: default policy, known at compile time)
        if constexpr (GROUP == 16 && ENV != 0) {
            {
                const float m = row16_fmax(fmaxf(fmaxf(fabsf(au[0]), fabsf(au[1])), fmaxf(fabsf(au[2]), fabsf(au[3]))));
                // block 4 q + (lane >> 4) from the first lane of its row; the other lanes point past the range (dropped)
                const int voff = (lane & 15) == 0 ? (lane >> 4) * 4 : 0x40000000;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rs_env, voff, (int)q * 16, 0);
            }
        }
    };

    // ---- the pipeline, once per channel of this workgroup ----
    // Audio of pass q is computed under the matrix stage of pass q+1 and stored right behind the mix stage of
    // pass q+2 -- in FRONT of that pass's prefetch loads: loads and stores share one in-order counter (vmcnt),
    // so a store issued shortly before loaded data is consumed makes the wave wait for the write acknowledge.
    // The last pass of a channel prefetches the first pass and the state of the workgroup's next channel, so a
    // channel switch costs no cold memory round trip and the Toeplitz fragments are loaded once per workgroup.
    for (;;) {
        install_state();
        float au[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        lds_order();
        if (DEC2 ? cur_out(0) * MO != (uint32_t)G::T : cur_out(0) != G::P) mix(0, std::true_type{});       // (a pass that does not fill the tile's inputs: a call's tail, 240- / 192-output passes)
        else mix(0, std::false_type{});
        prefetch(1);
        lds_order();
        {
            u4v cb[NCB];
            mfma_phase([](int) {});
            cb_read(cb);
            lds_order();
            cb_write(cb);
            dwrite();
        }
        lds_order();
        for (uint32_t pass = 1; pass < npass; ++pass) {
            if (DEC2 ? cur_out(pass) * MO != (uint32_t)G::T : cur_out(pass) != G::P) mix(pass, std::true_type{});
            else mix(pass, std::false_type{});
            store_audio(pass - 2, au);                                // pass 1: offset -1 pass = out of range, dropped
            prefetch(pass + 1);
            lds_order();
            u4v cb[NCB];
            v4f dt = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (keeps) dt = *reinterpret_cast<const v4f *>(D + dt_off + pq);      // (the pass before this one was a full one)
            gd.thr = thr_prev; gd.thr_h = fmaxf(thr_prev, thr_pp);    // the demodulator below belongs to the pass before
            gd.hist = pass == 1 ? hist_mask : 0ull;
            const int nvb_full = GROUP == 0 ? (int)(pq / (4u * (uint32_t)group)) : 64;
            mfma_phase([&](int kk) { demod_piece(kk, au, nvb_full); });   // matrix pipe over the vector work of the pass before
            cb_read(cb);
            lds_order();
            cb_write(cb);
            if (keeps) *reinterpret_cast<v4f *>(D + dt_off) = dt;
            dwrite();
            lds_order();
        }
        store_audio(npass - 2, au);
        load_state(c_nx);                                             // the next channel's state, under this channel's last demodulator pass
        gd.thr = thr_cur; gd.thr_h = fmaxf(thr_cur, thr_prev);
        gd.hist = npass == 1 ? hist_mask : 0ull;
        demod(au, (int)(tail_out / (4u * (uint32_t)group)));          // DSP blocks of the last pass that exist
        store_audio(npass - 1, au);

        // ---- parity guard: count (per-channel words: no atomics); SELENITE_ARITH_AUTO: a guarded channel keeps its pre-call
        // state and raises its rerun flag (the flag of every channel is rewritten every call) ----
        const bool keep_state = gd.n != 0u && p.rerun_flag != nullptr;   // wave-uniform
        if (lane == 0) {
            if (gd.n != 0u && p.guard_ch) { p.guard_ch[c] = sat_add_u32(p.guard_ch[c], gd.n); p.guard_calls[c] = sat_add_u32(p.guard_calls[c], 1u); }
            // handover blocks the rerun cannot repair: the call before stayed on the matrix kernel and left no hist_ext (a short call)
            // (or left them but the repair has been switched off since)
            if (keeps && gd.nh != 0u && (prev_prov == kProvSplit || (prev_prov == kProvSplitExt && !p.hist_ext)) && p.guard_hand) p.guard_hand[c] = sat_add_u32(p.guard_hand[c], gd.nh);      // (AM reads no history)
            if (p.rerun_flag) {
                const uint32_t kept = !keeps ? (((prev_prov == kProvExact ? kProvExact : kProvSplit) << kProvShift) | (prev_buf << kExtBufShift))      // (no samples: the format bit is void)
                                              : (((ext_on ? kProvSplitExt : kProvSplit) << kProvShift) | ((prev_buf ^ 1u) << kExtBufShift) |
                                                 (ext_on && sizeof(TIn) == 2 ? kExtQ15 : 0u));
                // (kept on the matrix kernel: the level of the last pass goes with the state -- the first blocks of the next call read
                // Hilbert-pair history computed from those samples; rounded up to the 24 bits the word has room for)
                p.rerun_flag[c] = keep_state ? (kFlagRerun | (st_word_cur & (kExtQ15 | (kProvMask << kProvShift) | (1u << kExtBufShift))))
                                             : (kept | ((b_last + 0xFFu) & kLvlMask));
            }
        }
        // ---- streaming state of the channel back to HBM (exact f32) ----
        lds_order();
        if (!keep_state) {
            const int ndr = (int)p.nd, Fr = G::HQ4 * M + 1 - ndr;
            float *stI = p.dec_state + (size_t)c * 2 * (ndr - 1), *stQ = stI + (ndr - 1);
#pragma unroll
            for (int j = 0; j < GS::HS / kWave; ++j) {
                const int s = j * kWave + lane - Fr;
                const v2f h = Hf[j * kWave + lane];
                if (s >= 0) { st_st(stI + s, h.x); st_st(stQ + s, h.y); }
            }
        }
        if (keeps) {                                                  // AM never ran the Hilbert pair: its state stays (FM keeps the delay lines running)
            int lane_o = lane;                                        // (opaque: the per-lane 64-bit offsets of this loop, hoisted out of the channel
            asm volatile("" : "+v"(lane_o));                          // loop, were the kernel's only scratch -- and a kernel with scratch pays ~12 us per launch)
            if (!keep_state)
                for (int i = lane_o; i < 2 * G::HH4; i += kWave) {
                    const int rail = i / G::HH4, mi = i % G::HH4, s = mi - G::FH;
                    if (s >= 0) st_st(p.fir_state + ((size_t)c * 2 + rail) * G::HH + s, D[rail * G::DLEN + tail_out + mi]);
                }
        }
        if (lane == 0 && !keep_state) {
            if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
            if (p.agc) p.gain[c] = gain;
        }
        c = c_nx;
        if (c >= p.channels) break;
        c_nx = ch_next();
        lds_order();                                                  // the state reads above before the next channel's installs
        rs_in = rs_in_next;
        rs_in_next = in_rsrc(c_nx);
        rs_out = make_rsrc(dst + (size_t)c * p.out_stride, p.nout * (W::kBytes / 4));
        if constexpr (ENV != 0) rs_env = env_rsrc(c);
    }
    if (nonfinite) p.flags[0] = 1u;                                   // ARM_MATH_NANINF, read by selenite_rx_sync
}

// ------------------------------------------------------------------------------------------
// k_hilb_split16<NCO, NH, TIn, TOut, AM> -- SELENITE_ARITH_SPLIT16 for the no-decimator shapes
// (BASELINE cfg1 / cfg2 / cfg5: M = 1, DSP block 256): the Hilbert FIR (arm_fir_f32.c:640-936) on the
// 16-bit matrix pipe.  Those shapes are VALU-bound by the Hilbert tap loop (64 non-zero taps of 127 per
// output); as a banded-Toeplitz product
//     D[i][m] = sum_k A[i][k] B[k][m],   A[i][k] = st[16 i + k],   B[k][m] = h[k - m]
// (st = [NH-1 history | 256 new samples] of the Q rail, 16 rows of 16 outputs, K = NH + 15) it is 3 MFMAs
// per k-step of 32 with the f16 hi/lo split and the block floating point of k_ssb_split16: the scale 2^s
// of a pass puts the largest |Q| of [history | new] into [2^14, 2^15); the Q history is kept in f32 beside
// the images and re-split when s changes.  15 v_mfma_f32_16x16x32_f16 per pass instead of 128 v_pk_fma +
// 64 v_readlane.  The I rail is a pure delay (unit-impulse FIR) and stays f32; the streaming state leaves
// from the f32 rails, bit-exact.  The MFMA result layout (lane holds outputs 64(l>>4) + 16 r + (l&15)) goes
// through a 1 KB LDS transpose so the audio leaves as one coalesced 16-byte store per lane.
// ------------------------------------------------------------------------------------------
template <int NCO, int NH, typename TIn, typename TOut, int AM = 0>
__global__ __launch_bounds__(64, 2) void k_hilb_split16(RxParams p, FusedArgs fa, const TIn *__restrict__ src,
                                                        TOut *__restrict__ dst)
{
    using GH = GeoH<NH>;
    using R = BRaw<TIn>;
    using W = BOut<TOut>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    // SELENITE_ARITH_AUTO in ONE launch (round 5; FusedArgs::inl).  Until round 4 a call was three launches: this kernel (which raises the rerun
    // bit of a channel it guards and leaves its state alone), k_hist_exact (the dense list of those channels) and the rerun pass of the
    // bit-exact kernel -- the last two empty in the steady state of a clean workload and still 4 us per call (1 % of a cfg5-sized call,
    // profiles/r5/ab_forms_hilb.txt).  One channel per workgroup: the workgroup that guarded the channel recomputes it ITSELF, with the body
    // of the bit-exact kernel (ssb_fused_body as the rerun pass: same guard evaluation, same hysteresis, same words), behind the matrix pass
    // (a channel it has just flagged) or instead of it (a channel the exact arithmetic holds) -- word_t != 0.  Which arithmetic serves a
    // channel is decided by its word alone, so both forms give the same bits; the load stays even by construction (the grid IS the channels).
    // (The persistent k_ssb_split16 keeps the three launches: its workgroups share channels statically, a dense list evens the recomputation
    // out -- and the bit-exact body inlined behind its pass pipeline took the registers the pipeline has to spare: 12 bytes of scratch or
    // four more v_readlane per pass, +0.6 % on the raw kernel for 0.2 % off the AUTO call, profiles/r5/ab_forms.txt.)
    constexpr bool INLT = AM == 0;
    uint32_t word_t = 0u;
    do {                                                              // (the matrix pass: left by `break`)
    // SELENITE_ARITH_AUTO, hysteresis (round 4): a channel the exact kernel holds is its alone -- the rerun pass of this call serves it
    if (p.rerun_flag != nullptr) {
        const uint32_t w0 = p.rerun_flag[c];
        if ((w0 & kFlagHold) != 0u) {                                 // wave-uniform
            if (INLT && fa.inl != 0u) word_t = w0;
            break;
        }
    }
    float *tab = lds + GH::oTab;
    _Float16 *Xh = reinterpret_cast<_Float16 *>(lds + GH::oX), *Xl = Xh + GH::IMG;
    float *dI = lds + GH::oDI, *dQ = lds + GH::oDQ, *O = lds + GH::oO;      // f32 rails [HH history | 256 new]
    // a pass produces pq audio samples: 256, or (round 4) the largest whole number of DSP blocks in 256 when the block does not divide
    // it -- 192 for DSP blocks of 192 frames, BASELINE cfg2's literal 48 000 samples = 250 of them.  The 256-sample
    // tile is always mixed and multiplied in full (input beyond the call reads as zeros); with pq < 256 its last outputs belong to
    // the next pass and are dropped by the store's lane mask, and the histories take the samples in front of pq.
    // (round 4, late) the LAST pass of a call may be shorter (a call is a whole number of DSP blocks, not of passes -- cfg2's 48 000
    // samples in DSP blocks of 128 are 187 passes of 256 and one of 128): its missing input reads as zeros and its surplus audio is
    // dropped by the buffer range checks, the AGC walks the blocks that exist, the state is taken behind the last sample that exists.
    const uint32_t pq = fa.pass_out;
    const uint32_t npass = (p.nout + pq - 1u) / pq;
    const uint32_t tail_out = p.nout - (npass - 1u) * pq;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(src + (size_t)c * p.in_stride * 2, p.block_size * (R::kBytes / 2));
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(dst + (size_t)c * p.out_stride, p.nout * (W::kBytes / 4));
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(p.lo, NCO == 2 ? p.block_size * 8u : 0u);
    const int kInPass = (int)pq * (R::kBytes / 2);
    typename R::type raw[2];
    u4v lo4[2];
    auto prefetch = [&](uint32_t pass) {                              // pass == npass: out of range, zeros, no traffic
        const int so = pass < npass ? (int)pass * kInPass : (int)(p.block_size * (R::kBytes / 2));
        // (pq < 256: the samples behind pq feed only outputs that are dropped -- not loaded: out of range, zeros, no traffic; they are
        // the next pass's to load)
#pragma unroll
        for (int i = 0; i < 2; ++i) raw[i] = R::load(rs_in, (uint32_t)(128 * i + 2 * lane) < pq ? lane * R::kBytes + i * 64 * R::kBytes : 0x40000000, so);
        if constexpr (NCO == 2) {
            const int sl = pass < npass ? (int)(pass * pq) * 8 : (int)p.block_size * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) lo4[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, (uint32_t)(128 * i + 2 * lane) < pq ? lane * 16 + i * 1024 : 0x40000000, sl, 0);
        }
    };
    prefetch(0);

    h8 Bh[GH::KS], Bl[GH::KS];
    {
        const h8 *bt = static_cast<const h8 *>(fa.btab16);
#pragma unroll
        for (int kk = 0; kk < GH::KS; ++kk) {
            Bh[kk] = bt[(2 * kk + 0) * 64 + lane];
            Bl[kk] = bt[(2 * kk + 1) * 64 + lane];
        }
    }
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // image slots u (even), u + 1 of the Q rail <- two samples times the block scale
    auto put = [&](int u, float x0, float x1, float pre) {
        const v2f s = v2f{ x0, x1 } * v2f{ pre, pre };
        const h2 h = __builtin_convertvector(s, h2);
        const h2 l = __builtin_convertvector(s - __builtin_convertvector(h, v2f), h2);
        const int ph = GH::phys(u);
        *reinterpret_cast<h2 *>(Xh + ph) = h;
        *reinterpret_cast<h2 *>(Xl + ph) = l;
    };
    // the read-only slack behind the samples meets zero taps only, but must hold finite numbers
    for (int u = GH::HH + 256 + 2 * lane; u < GH::XN; u += 2 * kWave) put(u, 0.0f, 0.0f, 1.0f);
    // state: both rails' histories in f32 (branch-free: lanes beyond the history repeat its last pair)
    const int hv = 2 * lane < GH::HH ? 2 * lane : GH::HH - 2;       // this lane's history pair
    uint32_t b_hist;                                                  // bit pattern of the largest |Q| of the history
    {
        const float *stI = p.fir_state + (size_t)c * 2 * GH::HH, *stQ = stI + GH::HH;
        const float i0 = stI[hv], i1 = stI[hv + 1], q0 = stQ[hv], q1 = stQ[hv + 1];
        *reinterpret_cast<float2 *>(dI + hv) = make_float2(i0, i1);
        *reinterpret_cast<float2 *>(dQ + hv) = make_float2(q0, q1);
        b_hist = wave_umax_bits_dpp(fmaxf(fabsf(q0), fabsf(q1)));
    }
    GuardPass gd{ 0.0f, 1ull, 0u };                                   // parity guard: one DSP block per pass ...
    const int group = (int)fa.group;
    if (pq != 256u || group != 64) {                                  // ... or pq / (4 group) of them (run-time geometry)
        gd.first = 0ull;
        for (int l = 0; l < 64; l += group) gd.first |= 1ull << l;
    }
    const uint32_t ph0 = NCO ? p.phase[c] : 0u, step = NCO ? p.step[c] : 0u;
    float gain = p.gain[c];
    const int mcol = lane & 15, rg = lane >> 4;
    int s_cur = 0x7fff;
    bool nonfinite = false;                                           // any audio sample NaN / Inf (x * 0 is NaN iff x is)
    lds_order();

    // (the pass as a function of its length: the loop over the whole passes calls it with pq -- the code of round 3, instruction for
    // instruction --, the partial last pass is a second copy of it with its own length)
    auto one_pass = [&](uint32_t pass, uint32_t cur) {               // cur: audio samples of this pass (whole DSP blocks)
        const uint32_t n0 = pass * pq;
        const int nvb = (int)(cur / (4u * (uint32_t)group));
        // ---- 1. NCO mix; both rails f32 into LDS; Q rail split into the f16 images at the block scale ----
        v2f ma[2], mb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v2f a, b;
            R::unpack(raw[i], a, b);
            if constexpr (NCO == 2) {
                cmul_pk2(a, b, v2f{ __uint_as_float(lo4[i].x), __uint_as_float(lo4[i].y) },
                         v2f{ __uint_as_float(lo4[i].z), __uint_as_float(lo4[i].w) }, ma[i], mb[i]);
            } else if constexpr (NCO == 1) {
                const uint32_t pe = ph0 + 2u * lane * step + (n0 + 128u * i) * step;
                v2f la, lb;
                nco_lo_pair(tab, pe, pe + step, la, lb);
                cmul_pk2(a, b, la, lb, ma[i], mb[i]);
            } else {
                ma[i] = a;
                mb[i] = b;
            }
        }
        prefetch(pass + 1);
        float au[4], m_pre = -1.0f;                                   // m_pre: this lane's max |audio| in the matrix layout (the SSB modes)
        if constexpr (AM != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<float2 *>(O + 128 * i + 2 * lane) = make_float2(cmag<0>(ma[i].x, ma[i].y), cmag<0>(mb[i].x, mb[i].y));
            lds_order();
        } else {
            // block exponent: the largest |Q| of the new samples and of the history; the new samples that will be
            // the NEXT pass's history (the last HH of the pass) are tracked on the side
            float mq = 0.0f, mt = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = 128 * i + 2 * lane;
                const float m2 = fmaxf(fabsf(ma[i].y), fabsf(mb[i].y));
                mq = fmaxf(mq, m2);
                mt = fmaxf(mt, (n >= (int)pq - GH::HH || pq != 256u) ? m2 : 0.0f);   // HH is even: a pair is inside or outside as a whole (pq < 256: the whole tile, a safe bound)
            }
            const uint32_t b_tail = wave_umax_bits_dpp(mt);
            const uint32_t b_need = max(max(wave_umax_bits_dpp(mq), b_tail), b_hist);  // the largest |Q| the matrix product sees
            const uint32_t e_need = b_need >> 23;
            gd.thr = __uint_as_float(b_need) * p.guard_ratio;
            int s_new = 141 - (int)e_need;
            s_new = s_new > 127 ? 127 : (s_new < -126 ? -126 : s_new);
            if (s_new != s_cur) {                                     // wave-uniform; always in the first pass
                const float2 hq = *reinterpret_cast<const float2 *>(dQ + hv);
                put(hv, hq.x, hq.y, __uint_as_float((uint32_t)(s_new + 127) << 23));
                s_cur = s_new;
            }
            b_hist = b_tail;
            const float pre = __uint_as_float((uint32_t)(s_cur + 127) << 23);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = 128 * i + 2 * lane;
                *reinterpret_cast<float2 *>(dI + GH::HH + n) = make_float2(ma[i].x, mb[i].x);
                *reinterpret_cast<float2 *>(dQ + GH::HH + n) = make_float2(ma[i].y, mb[i].y);
                put(GH::HH + n, ma[i].y, mb[i].y, pre);
            }
            lds_order();
            // ---- 2. Hilbert FIR of the Q rail: 3 f16 MFMAs per k-step (small terms first, one accumulator) ----
            v4f acc = { 0.0f, 0.0f, 0.0f, 0.0f };
            float i2[4];                                                  // the delayed I rail of this lane's four outputs, read while the matrix pipe works
#pragma unroll                                                            // (behind the product each read would wait for the O[] store in front of it)
            for (int r = 0; r < 4; ++r) i2[r] = dI[64 * rg + 16 * r + mcol + fa.delay_idx] + 0.0f;
#pragma unroll
            for (int kk = 0; kk < GH::KS; ++kk) {
                const int u = 16 * mcol + 8 * rg + 32 * kk;                // A[i = l&15][k = 32kk + 8(l>>4) ..+7] = st[16 i + k]
                const int ph = GH::phys(u);
                const h8 ah = *reinterpret_cast<const h8 *>(Xh + ph), al = *reinterpret_cast<const h8 *>(Xl + ph);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bl[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, Bh[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bh[kk], acc, 0, 0, 0);
            }
            // ---- 3. delay on I, sideband combine; transpose through LDS ----
            const int ex = -(s_cur + fa.split_sc);
            float m_mx = 0.0f;
            float q2[4];                                                  // -+ the Hilbert rail: the sign rides on v_ldexp_f32's input modifier, the sideband is a
            if (fa.upper) {                                               // wave-uniform BRANCH (the empty asm cannot be speculated): four selects less per pass
#pragma unroll
                for (int r = 0; r < 4; ++r) q2[r] = __builtin_ldexpf(-acc[r], ex);
                asm volatile("" : "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]), "+v"(q2[3]));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) q2[r] = __builtin_ldexpf(acc[r], ex);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 64 * rg + 16 * r + mcol;                     // D[row 4 rg + r][col mcol]
                const float o = i2[r] + q2[r];                             // (i2 - q == i2 + (-q) bit for bit)
                O[n] = o;
                m_mx = fmaxf(m_mx, fabsf(o));
            }
            m_pre = m_mx;
            lds_order();
        }
        // ---- 4.-5. AGC on the DSP block (= the pass), coalesced store ----
        {
            const float4 o4 = lds_ld4f(O + 4 * lane);
            au[0] = o4.x; au[1] = o4.y; au[2] = o4.z; au[3] = o4.w;
        }
        if (pq == 256u && group == 64) agc_pass<64>(p.agcp, p.agc, lane, 64, au, gain, 1, gd, m_pre);   // (AM: exact arithmetic, thr stays 0: never guarded)
        else if (pq == 256u && group == 32) agc_pass<32>(p.agcp, p.agc, lane, 32, au, gain, nvb, gd, m_pre);   // DSP blocks of 128 frames (cfg2 literally): DPP, no LDS round trips
        else agc_pass<0>(p.agcp, p.agc, lane, group, au, gain, nvb, gd);
        {
            const float z = __builtin_fmaf(au[3], 0.0f, __builtin_fmaf(au[2], 0.0f, __builtin_fmaf(au[1], 0.0f, au[0] * 0.0f)));
            nonfinite = nonfinite || (z != z);
        }
        W::store(rs_out, (uint32_t)lane < pq / 4u ? lane * W::kBytes : 0x40000000, (int)(pass * pq) * (W::kBytes / 4), au, false, p.q15_round);
        // ---- 6. history: last NH-1 samples of both f32 rails and of both images to the front ----
        if constexpr (AM == 0) {
            const float2 ti = *reinterpret_cast<const float2 *>(dI + cur + hv);
            const float2 tq = *reinterpret_cast<const float2 *>(dQ + cur + hv);
            const uint32_t th = *reinterpret_cast<const uint32_t *>(Xh + GH::phys((int)cur + hv));
            const uint32_t tl = *reinterpret_cast<const uint32_t *>(Xl + GH::phys((int)cur + hv));
            lds_order();
            *reinterpret_cast<float2 *>(dI + hv) = ti;
            *reinterpret_cast<float2 *>(dQ + hv) = tq;
            *reinterpret_cast<uint32_t *>(Xh + GH::phys(hv)) = th;
            *reinterpret_cast<uint32_t *>(Xl + GH::phys(hv)) = tl;
        }
        lds_order();
    };
    const uint32_t nfull = tail_out == pq ? npass : npass - 1u;
    for (uint32_t pass = 0; pass < nfull; ++pass) one_pass(pass, pq);
    if (nfull != npass) one_pass(nfull, tail_out);
    // ---- parity guard: count (per-channel words: no atomics); SELENITE_ARITH_AUTO: a guarded channel keeps its pre-call state
    // and raises its rerun flag (the flag of every channel is rewritten every call) ----
    {
        const bool keep_state = gd.n != 0u && p.rerun_flag != nullptr;   // wave-uniform
        if (lane == 0) {
            if (gd.n != 0u && p.guard_ch) { p.guard_ch[c] = sat_add_u32(p.guard_ch[c], gd.n); p.guard_calls[c] = sat_add_u32(p.guard_calls[c], 1u); }
            if (p.rerun_flag) p.rerun_flag[c] = keep_state ? 1u : 0u;
        }
        if (keep_state) {
            if (nonfinite) p.flags[0] = 1u;
            if (INLT && fa.inl != 0u) word_t = kFlagRerun;            // (the word just written)
            break;
        }
    }
    // ---- epilogue: arm_fir_f32 pState tails (the last NH-1 samples of each rail), exact f32 ----
    if constexpr (AM == 0) {
        if (2 * lane < GH::HH) {
            float *stI = p.fir_state + (size_t)c * 2 * GH::HH, *stQ = stI + GH::HH;
            const float2 ti = *reinterpret_cast<const float2 *>(dI + hv), tq = *reinterpret_cast<const float2 *>(dQ + hv);
            stI[hv] = ti.x; stI[hv + 1] = ti.y;
            stQ[hv] = tq.x; stQ[hv + 1] = tq.y;
        }
    }
    if (nonfinite) p.flags[0] = 1u;                                   // ARM_MATH_NANINF, read by selenite_rx_sync
    if (lane == 0) {
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
        if (p.agc) p.gain[c] = gain;
    }
    } while (0);
    if constexpr (INLT) {
        if (word_t != 0u) {                                           // wave-uniform
            RxParams p2 = p;                                          // the rerun pass's view (rx_fused.hip: launch_shape, rerun())
            p2.chan_flags = p.rerun_flag;
            p2.rerun_flag = nullptr;
            p2.chan_list = nullptr; p2.chan_count = nullptr; p2.chan_count_next = nullptr; p2.rerun_seen = nullptr;
            FusedArgs fa2 = fa;
            fa2.nco = p.nco == 2 ? 2u : (p.nco == 1 ? ((p.lo_period == 256 && fa.pass_out == 256) ? 4u : 1u) : 0u);      // (launch_one)
            __syncthreads();
            ssb_fused_body<0, 0, 1, NH, TIn, TOut, 0, true>(p2, fa2, src, dst, FusedInl{ c, word_t });
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side: dispatch over the instantiated shapes
// ------------------------------------------------------------------------------------------
template <int NCO, int ND, int M, int NH, typename TIn, typename TOut, int AM, int GROUP, int FLAVOUR = 0>
static hipError_t launch_k(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    using GS = GeoS<NCO, ND, M, NH>;
    constexpr size_t lds = (size_t)GS::total * sizeof(float);
    static_assert(lds <= 48 * 1024, "k_ssb_split16 LDS image");
    if (fa.inl) return hipErrorNotSupported;                      // (the one-launch form of SELENITE_ARITH_AUTO: k_hilb_split16 only)
    // persistent grid: as many single-wave workgroups as the device keeps resident, each running channels
    // b, b + grid, b + 2 grid, ...  (SELENITE_RX_SPLIT16_GRID=0: one workgroup per channel, the round-1 launch shape)
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ssb_split16<NCO, ND, M, NH, TIn, TOut, AM, GROUP, FLAVOUR>, 64, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu <= 0)
            resident = -1;
        else
            resident = per_cu * prop.multiProcessorCount;
        if (const char *e = diag_env("SELENITE_RX_SPLIT16_GRID")) resident = std::atoi(e) > 0 ? std::atoi(e) : -1;
    }
    const uint32_t grid = resident > 0 && (uint32_t)resident < p.channels ? (uint32_t)resident : p.channels;
    if constexpr (GROUP == 16 && AM == 0 && sizeof(TOut) == 4) {
        if (p.env_part) {                         // global gain, phase 1: the flavour that also leaves the block maxima behind
            hipLaunchKernelGGL((k_ssb_split16<NCO, ND, M, NH, TIn, TOut, AM, GROUP, 1>), dim3(grid), dim3(64), lds, st, p, fa,
                               static_cast<const TIn *>(src), static_cast<TOut *>(dst));
            return hipGetLastError();
        }
    }
    if (p.env_part) return hipErrorNotSupported;  // the host side asks for the maxima only from launches that provide them
    hipLaunchKernelGGL((k_ssb_split16<NCO, ND, M, NH, TIn, TOut, AM, GROUP, FLAVOUR>), dim3(grid), dim3(64), lds, st, p, fa,
                       static_cast<const TIn *>(src), static_cast<TOut *>(dst));
    return hipGetLastError();
}

// DSP-block geometry: 16 / 64 lanes per block (block = 256 / 1024 inputs) get the DPP reductions, any
// other power of two the run-time variant; AM and the NCO-less chain (rare) only the run-time variant
template <int NCO, int ND, int M, int NH, typename TIn, typename TOut>
static hipError_t launch_io(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    if constexpr (M == 4) {
        if (fa.am && fa.dec2) return launch_k<NCO, ND, M, NH, TIn, TOut, 1, 0, 2>(p, fa, src, dst, st);
    }
    if (fa.am && !fa.dec2) return launch_k<NCO, ND, M, NH, TIn, TOut, 1, 0>(p, fa, src, dst, st);
    if constexpr (M == 4) {
        if (fa.dec2) return launch_k<NCO, ND, M, NH, TIn, TOut, 0, 0, 2>(p, fa, src, dst, st);     // by 2 M on the by-M product: run-time geometry, its own instantiation
    }
    if (fa.dec2) return hipErrorNotSupported;
    if constexpr (NCO != 0 && M == 2) {
        if (fa.group == 32) return launch_k<NCO, ND, M, NH, TIn, TOut, 0, 32>(p, fa, src, dst, st);      // DSP block 256 inputs / 2
    } else if constexpr (NCO != 0) {
        if (fa.group == 16) return launch_k<NCO, ND, M, NH, TIn, TOut, 0, 16>(p, fa, src, dst, st);
        if (fa.group == 64) return launch_k<NCO, ND, M, NH, TIn, TOut, 0, 64>(p, fa, src, dst, st);
    }
    return launch_k<NCO, ND, M, NH, TIn, TOut, 0, 0>(p, fa, src, dst, st);
}

template <int ND, int M, int NH, typename T>
static hipError_t launch_nco(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    if constexpr (Geo<ND, M, NH>::T % 256 == 0) {
        if (p.nco == 2 && p.lo_period == 256) return launch_io<3, ND, M, NH, T, T>(p, fa, src, dst, st);      // LO held in registers
        if (p.nco == 1 && p.lo_period == 256) return launch_io<4, ND, M, NH, T, T>(p, fa, src, dst, st);      // every channel its own LO on the fs / 256 grid: in registers, computed once per channel
    }
    if (p.nco == 2) return launch_io<2, ND, M, NH, T, T>(p, fa, src, dst, st);
    if (p.nco == 1) return launch_io<1, ND, M, NH, T, T>(p, fa, src, dst, st);
    return launch_io<0, ND, M, NH, T, T>(p, fa, src, dst, st);
}

// the int16-slot instantiations of k_ssb_split16 live in rx_split16_q15.hip (a translation unit of its own: compiles in parallel)
hipError_t launch_ssb_split16_q15(int nd, int m, int nh, const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st);

}  // namespace srx
