/*
 * rx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rx_oracle.h).
 *
 * Plain-C restatement of the CMSIS-DSP 1.5.3 f32 primitives on the Selenite RX block path and of
 * the build-defined chain that composes them.  Written fresh as straightforward loops with the
 * same per-output operation ORDER as the reference; compiled with
 *   gcc -O2 -ffp-contract=off   (no -ffast-math, no -march=native)
 * so that no multiply-add is fused unless this file asks for it with fmaf().
 *
 * All citations are relative to /root/reference/Drivers/CMSIS/DSP/Source unless stated.
 */
#include "rx_oracle.h"
#include "fm_atan.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ARITH_FMA(a) ((a) == SELENITE_ARITH_FMA)

/* acc + x*c with the rounding the arithmetic mode prescribes */
static inline float mac(float acc, float x, float c, int arith)
{
    if (ARITH_FMA(arith))
        return fmaf(x, c, acc);
    float p = x * c;
    return acc + p;
}

/* ------------------------------------------------------------------------------------------
 * sinTable_f32[513]: CommonTables/arm_common_tables.c:21895.  The reference documents the
 * generator (:21881-21891, sin(2*pi*n/512)) and stores every entry as an 8-decimal literal, so
 * the table is float(round(sin(2*pi*n/512), 8 decimals)).  Regenerated here; tests compare all
 * 513 entries with the reference table (and with tests/golden/sintable_f32.bin on the GPU box).
 * ------------------------------------------------------------------------------------------ */
static float g_sin_table[513];
static pthread_once_t g_sin_once = PTHREAD_ONCE_INIT;

static void sin_table_build(void)
{
    for (int n = 0; n <= 512; ++n) {
        double s = sin(2.0 * 3.14159265358979323846 * (double)n / 512.0);
        double r = nearbyint(s * 1e8);           /* 8 decimals, as printed in the reference */
        char buf[32];
        /* go through the decimal literal so the float is the correctly rounded value of the
         * same text the reference source holds */
        long long q = (long long)r;
        int neg = signbit(s) != 0;               /* entry 512 is printed as -0.00000000f */
        if (q < 0) q = -q;
        int len = 0;
        buf[len++] = neg ? '-' : '+';
        buf[len++] = (char)('0' + (int)(q / 100000000LL));
        buf[len++] = '.';
        long long frac = q % 100000000LL;
        for (long long d = 10000000LL; d >= 1; d /= 10) {
            buf[len++] = (char)('0' + (int)(frac / d));
            frac %= d;
        }
        buf[len] = 0;
        g_sin_table[n] = strtof(buf, NULL);
    }
}

const float *orc_sin_table(void)
{
    pthread_once(&g_sin_once, sin_table_build);
    return g_sin_table;
}

/* FastMathFunctions/arm_sin_f32.c:72-119 */
float orc_sin_f32(float x, int arith)
{
    const float *T = orc_sin_table();
    if ((x < 0.0f) && (x >= -1.9e-7f))          /* :82-84 small-negative shortcut */
        return x;
    float in = x * 0.159154943092f;             /* :88 */
    int32_t n = (int32_t)in;                    /* :91 */
    if (x < 0.0f)                               /* :94-97 (tests x, not in) */
        n--;
    in = in - (float)n;                         /* :100 */
    float findex = 512.0f * in;                 /* :103 */
    uint16_t index = ((uint16_t)findex) & 0x1ff;/* :105 */
    float fract = findex - (float)index;        /* :108 */
    float a = T[index], b = T[index + 1];       /* :111-112 */
    float w = 1.0f - fract;
    (void)arith;                                /* FMA mode fuses FIR tap loops only */
    float p0 = w * a, p1 = fract * b;           /* :115 */
    return p0 + p1;
}

/* FastMathFunctions/arm_cos_f32.c:70-111 */
float orc_cos_f32(float x, int arith)
{
    const float *T = orc_sin_table();
    float p = x * 0.159154943092f;              /* :81 */
    float in = p + 0.25f;
    (void)arith;                                /* FMA mode fuses FIR tap loops only */
    int32_t n = (int32_t)in;                    /* :84 */
    if (in < 0.0f)                              /* :87-90 (tests in) */
        n--;
    in = in - (float)n;                         /* :93 */
    float findex = 512.0f * in;                 /* :96 */
    uint16_t index = ((uint16_t)findex) & 0x1ff;/* :97 */
    float fract = findex - (float)index;        /* :100 */
    float a = T[index], b = T[index + 1];       /* :103-104 */
    float w = 1.0f - fract;
    float p0 = w * a, p1 = fract * b;           /* :107 */
    return p0 + p1;
}

/* LO samples for an array of integer phases: lo[2n] = cos(x), lo[2n+1] = -sin(x), x = (float)(phase >> 8) * ORC_NCO_K
 * (the NCO of the chain below, step 1, exposed for the tests of the GPU kernels' restated sin/cos sequence) */
void orc_nco_lo(const uint32_t *phase, uint32_t n, float *lo)
{
    for (uint32_t k = 0; k < n; ++k) {
        float x = (float)(phase[k] >> 8) * ORC_NCO_K;
        lo[2 * k] = orc_cos_f32(x, SELENITE_ARITH_CMSIS);
        lo[2 * k + 1] = -orc_sin_f32(x, SELENITE_ARITH_CMSIS);
    }
}

/* ComplexMathFunctions/arm_cmplx_mult_cmplx_f32.c:72-192 (values per :186-187) */
void orc_cmplx_mult_cmplx_f32(const float *A, const float *B, float *dst, uint32_t n, int arith)
{
    for (uint32_t i = 0; i < n; ++i) {
        float a = A[2 * i], b = A[2 * i + 1], c = B[2 * i], d = B[2 * i + 1];
        float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
        (void)arith;                            /* FMA mode fuses FIR tap loops only */
        dst[2 * i]     = ac - bd;
        dst[2 * i + 1] = ad + bc;
    }
}

/* ComplexMathFunctions/arm_cmplx_conj_f32.c:71-166: real part copied, imaginary part negated */
void orc_cmplx_conj_f32(const float *src, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) {
        dst[2 * i] = src[2 * i];
        dst[2 * i + 1] = -src[2 * i + 1];
    }
}

/* the build-defined arctangent of the FM discriminator (fm_atan.h; CMSIS-DSP 1.5.3 has none) */
float orc_fm_atan2_f32(float y, float x) { return fm_atan2_f32(y, x); }

/* ComplexMathFunctions/arm_cmplx_mag_f32.c:72-149; arm_sqrt_f32 = sqrtf for in >= 0
 * (Include/arm_math.h:5726-5752) */
void orc_cmplx_mag_f32(const float *src, float *dst, uint32_t n, int arith)
{
    for (uint32_t i = 0; i < n; ++i) {
        float re = src[2 * i], im = src[2 * i + 1], s;
        float rr = re * re, ii = im * im;
        (void)arith;                            /* FMA mode fuses FIR tap loops only */
        s = rr + ii;
        dst[i] = (s >= 0.0f) ? sqrtf(s) : 0.0f;
    }
}

/* FilteringFunctions/arm_fir_decimate_f32.c:129-508.
 * state = [numTaps-1 history | block new samples]; output j = sum_k coeffs[k]*state[j*M+k],
 * ONE accumulator from 0.0f, k ascending (:193-260 / :265-284 / :441-455); the reference's
 * 4-output unroll runs across outputs only.  History copy-back :396-426. */
void orc_fir_decimate_f32(const float *coeffs, uint32_t num_taps, uint32_t M, float *state,
                          const float *src, float *dst, uint32_t block, int arith)
{
    float *cur = state + (num_taps - 1u);
    memcpy(cur, src, (size_t)block * sizeof(float));
    uint32_t nout = block / M;
    for (uint32_t j = 0; j < nout; ++j) {
        const float *px = state + (size_t)j * M;
        float acc = 0.0f;
        for (uint32_t k = 0; k < num_taps; ++k)
            acc = mac(acc, px[k], coeffs[k], arith);
        dst[j] = acc;
    }
    memmove(state, state + (size_t)nout * M, (size_t)(num_taps - 1u) * sizeof(float));
}

/* FilteringFunctions/arm_fir_f32.c:553-979 (the non-CM7, non-CM0 variant the firmware's
 * ARM_MATH_CM4 selects).  y[n] = sum_k coeffs[k]*state[n+k], single accumulator, k ascending;
 * the 8-output unroll forms each product separately and then adds it (:640-676), i.e. plain
 * "acc += x*c" without fusion.  History copy-back :947-978. */
void orc_fir_f32(const float *coeffs, uint32_t num_taps, float *state,
                 const float *src, float *dst, uint32_t block, int arith)
{
    float *cur = state + (num_taps - 1u);
    memcpy(cur, src, (size_t)block * sizeof(float));
    for (uint32_t n = 0; n < block; ++n) {
        const float *px = state + n;
        float acc = 0.0f;
        for (uint32_t k = 0; k < num_taps; ++k)
            acc = mac(acc, px[k], coeffs[k], arith);
        dst[n] = acc;
    }
    memmove(state, state + block, (size_t)(num_taps - 1u) * sizeof(float));
}

/* FilteringFunctions/arm_biquad_cascade_df1_f32.c:165-407.
 * y = (b0*x) + (b1*x1) + (b2*x2) + (a1*y1) + (a2*y2) evaluated left to right (:220, :297),
 * feedback ADDED (:52-63); coeffs {b0,b1,b2,a1,a2}, state {x1,x2,y1,y2} per stage (:319-322);
 * the reference runs stage-outer over the block, which yields the same values as this
 * sample-by-sample form because each stage only consumes the previous stage's output. */
void orc_biquad_cascade_df1_f32(const float *coeffs, uint32_t stages, float *state,
                                const float *src, float *dst, uint32_t block, int arith)
{
    (void)arith;
    const float *in = src;
    for (uint32_t s = 0; s < stages; ++s) {
        const float b0 = coeffs[5 * s], b1 = coeffs[5 * s + 1], b2 = coeffs[5 * s + 2];
        const float a1 = coeffs[5 * s + 3], a2 = coeffs[5 * s + 4];
        float x1 = state[4 * s], x2 = state[4 * s + 1], y1 = state[4 * s + 2], y2 = state[4 * s + 3];
        for (uint32_t n = 0; n < block; ++n) {
            float x = in[n], y;
            /* the recurrence keeps the reference rounding in BOTH arithmetic modes: fusing it
             * moves a high-Q cascade by > 1e-5 relative (measured 1.5e-5 on cfg4) */
            float p0 = b0 * x, p1 = b1 * x1, p2 = b2 * x2, p3 = a1 * y1, p4 = a2 * y2;
            y = p0 + p1;
            y = y + p2;
            y = y + p3;
            y = y + p4;
            dst[n] = y;
            x2 = x1; x1 = x; y2 = y1; y1 = y;
        }
        state[4 * s] = x1; state[4 * s + 1] = x2; state[4 * s + 2] = y1; state[4 * s + 3] = y2;
        in = dst;                                /* :324-329 later stages run in place */
    }
}

/* BasicMathFunctions/arm_add_f32.c:61,129 / arm_sub_f32.c:62,129 */
void orc_add_f32(const float *a, const float *b, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = a[i] + b[i];
}
void orc_sub_f32(const float *a, const float *b, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = a[i] - b[i];
}
/* BasicMathFunctions/arm_abs_f32.c:63,144 (fabsf) */
void orc_abs_f32(const float *src, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = fabsf(src[i]);
}
/* StatisticsFunctions/arm_max_f32.c:58-166: strict '<' keeps the FIRST maximum's index */
void orc_max_f32(const float *src, uint32_t n, float *result, uint32_t *index)
{
    float out = src[0];
    uint32_t idx = 0;
    for (uint32_t i = 1; i < n; ++i)
        if (out < src[i]) { out = src[i]; idx = i; }
    *result = out;
    if (index) *index = idx;
}
/* BasicMathFunctions/arm_scale_f32.c:77,148 */
void orc_scale_f32(const float *src, float scale, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = src[i] * scale;
}
/* SupportFunctions/arm_q15_to_float.c:65-113 */
void orc_q15_to_float(const int16_t *src, float *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = (float)src[i] / 32768.0f;
}
/* SupportFunctions/arm_float_to_q15.c:64-122, ARM_MATH_ROUNDING undefined (firmware build,
 * .cproject:44): (q15_t)__SSAT((q31_t)(x * 32768.0f), 16): truncate toward zero, saturate */
static void float_to_q15(const float *src, int16_t *dst, uint32_t n, int rounding)
{
    for (uint32_t i = 0; i < n; ++i) {
        float v = src[i] * 32768.0f;
        if (rounding) v += v > 0.0f ? 0.5f : -0.5f;     /* arm_float_to_q15.c:90-101 (#ifdef ARM_MATH_ROUNDING) */
        int32_t q;
        if (v != v) q = 0;                              /* NaN: UB in C; the firmware's VCVT.S32.F32 (and v_cvt_i32_f32) give 0 */
        else if (v >= 2147483648.0f) q = INT32_MAX;     /* out of int32 range is UB in C too: the FPU saturates */
        else if (v <= -2147483648.0f) q = INT32_MIN;
        else q = (int32_t)v;
        if (q > 32767) q = 32767;
        if (q < -32768) q = -32768;
        dst[i] = (int16_t)q;
    }
}
void orc_float_to_q15(const float *src, int16_t *dst, uint32_t n) { float_to_q15(src, dst, n, 0); }
/* the same function built with ARM_MATH_ROUNDING (selenite_rx_config::q15_rounding = 1): +-0.5 in float before the truncation */
void orc_float_to_q15_rounding(const float *src, int16_t *dst, uint32_t n) { float_to_q15(src, dst, n, 1); }

/* AGC gain law -- build-defined (DESIGN.md "AGC"): per DSP block,
 *   e = max(env, floor); d = clamp(target / e, gmin, gmax);
 *   g += (d < g ? attack : decay) * (d - g)                                                   */
float orc_agc_update(const selenite_rx_config *cfg, float gain, float env, int arith)
{
    float e = (env < cfg->agc_env_floor) ? cfg->agc_env_floor : env;
    float d = cfg->agc_target / e;
    if (d > cfg->agc_gain_max) d = cfg->agc_gain_max;
    if (d < cfg->agc_gain_min) d = cfg->agc_gain_min;
    float diff = d - gain;
    float rate = (diff < 0.0f) ? cfg->agc_attack : cfg->agc_decay;
    (void)arith;                                /* FMA mode fuses FIR tap loops only */
    float p = rate * diff;
    return gain + p;
}

/* ------------------------------------------------------------------------------------------
 * Chain (build-defined; DESIGN.md "Chain specification")
 * ------------------------------------------------------------------------------------------ */
struct orc_rx {
    selenite_rx_config cfg;
    float *dec_coeffs, *hilb_coeffs, *delay_coeffs, *biquad_coeffs;
    uint32_t *nco_step;
    /* per channel state */
    float *dec_state;    /* [C][2][nd-1 + block] */
    float *fir_state;    /* [C][2][nh-1 + block/M] */
    float *biq_state;    /* [C][stages][4] */
    float *gain;         /* [C] */
    uint32_t *phase;     /* [C] */
    size_t dec_stride, fir_stride;
};

/* FM (rxtx_if.h:41) keeps its one-sample memory in the delay lines of the FIR pair: it needs them */
static int mode_valid(uint8_t m, uint32_t nh_taps)
{
    if (m == SELENITE_MODE_FM) return nh_taps >= 2;
    return m == SELENITE_MODE_LSB || m == SELENITE_MODE_USB || m == SELENITE_MODE_CW ||
           m == SELENITE_MODE_CWR || m == SELENITE_MODE_AM || m == SELENITE_MODE_DIG ||
           m == SELENITE_MODE_PKT;
}

static float *dupf(const float *p, size_t n)
{
    if (!p || !n) return NULL;
    float *q = (float *)malloc(n * sizeof(float));
    memcpy(q, p, n * sizeof(float));
    return q;
}

int orc_rx_create(orc_rx **out, const selenite_rx_config *cfg)
{
    *out = NULL;
    if (!cfg || cfg->channels == 0 || cfg->block == 0 || cfg->decim == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (!mode_valid(cfg->mode, cfg->nh_taps)) return SELENITE_RX_ARGUMENT_ERROR;
    if (cfg->nd_taps == 0 && cfg->decim != 1) return SELENITE_RX_ARGUMENT_ERROR;
    if (cfg->nd_taps && !cfg->dec_coeffs) return SELENITE_RX_ARGUMENT_ERROR;
    if (cfg->nh_taps && (!cfg->hilb_coeffs || !cfg->delay_coeffs)) return SELENITE_RX_ARGUMENT_ERROR;
    if (cfg->n_biquad && !cfg->biquad_coeffs) return SELENITE_RX_ARGUMENT_ERROR;
    /* arm_fir_decimate_init_f32.c:74-97 */
    if (cfg->block % cfg->decim != 0) return SELENITE_RX_LENGTH_ERROR;

    orc_rx *S = (orc_rx *)calloc(1, sizeof(*S));
    S->cfg = *cfg;
    const uint32_t C = cfg->channels;
    S->dec_coeffs = dupf(cfg->dec_coeffs, cfg->nd_taps);
    S->hilb_coeffs = dupf(cfg->hilb_coeffs, cfg->nh_taps);
    S->delay_coeffs = dupf(cfg->delay_coeffs, cfg->nh_taps);
    S->biquad_coeffs = dupf(cfg->biquad_coeffs, 5u * cfg->n_biquad);
    S->nco_step = (uint32_t *)malloc(C * sizeof(uint32_t));
    for (uint32_t c = 0; c < C; ++c)
        S->nco_step[c] = cfg->nco_step ? cfg->nco_step[c] : cfg->nco_step_all;
    S->dec_stride = cfg->nd_taps ? (cfg->nd_taps - 1u + cfg->block) : 0;
    S->fir_stride = cfg->nh_taps ? (cfg->nh_taps - 1u + cfg->block / cfg->decim) : 0;
    S->dec_state = (float *)calloc((size_t)C * 2 * (S->dec_stride + 1), sizeof(float));
    S->fir_state = (float *)calloc((size_t)C * 2 * (S->fir_stride + 1), sizeof(float));
    S->biq_state = (float *)calloc((size_t)C * (4u * cfg->n_biquad + 1), sizeof(float));
    S->gain = (float *)malloc(C * sizeof(float));
    S->phase = (uint32_t *)calloc(C, sizeof(uint32_t));
    for (uint32_t c = 0; c < C; ++c) S->gain[c] = cfg->agc_gain_init;
    S->cfg.dec_coeffs = S->dec_coeffs; S->cfg.hilb_coeffs = S->hilb_coeffs;
    S->cfg.delay_coeffs = S->delay_coeffs; S->cfg.biquad_coeffs = S->biquad_coeffs;
    S->cfg.nco_step = S->nco_step;
    *out = S;
    return SELENITE_RX_SUCCESS;
}

void orc_rx_destroy(orc_rx *S)
{
    if (!S) return;
    free(S->dec_coeffs); free(S->hilb_coeffs); free(S->delay_coeffs); free(S->biquad_coeffs);
    free(S->nco_step); free(S->dec_state); free(S->fir_state); free(S->biq_state);
    free(S->gain); free(S->phase); free(S);
}

int orc_rx_set_mode(orc_rx *S, uint8_t mode)
{
    if (!mode_valid(mode, S->cfg.nh_taps)) return SELENITE_RX_ARGUMENT_ERROR;
    S->cfg.mode = mode;
    return SELENITE_RX_SUCCESS;
}

static int mode_is_cw(uint8_t m) { return m == SELENITE_MODE_CW || m == SELENITE_MODE_CWR; }
static int mode_is_upper(uint8_t m)
{
    return m == SELENITE_MODE_USB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_CW;
}

/* One channel, one DSP block, everything up to (not including) the AGC.
 * work: scratch of >= 8*block floats.  Returns max|audio| of the block in *env. */
static void chain_block(orc_rx *S, uint32_t c, const float *iq, float *audio, float *work, float *env)
{
    const selenite_rx_config *g = &S->cfg;
    const int ar = (int)g->arith;
    const uint32_t nb = g->block, M = g->decim, na = nb / M;
    float *mixed = work;            /* [nb][2] */
    float *lo = work + 2 * nb;      /* [nb][2] */
    float *ri = work + 4 * nb;      /* [nb] */
    float *rq = work + 5 * nb;      /* [nb] */
    float *di = work + 6 * nb;      /* [na] */
    float *dq = work + 7 * nb;      /* [na] */

    /* 1. NCO quadrature mix: LO = cos(x) - j sin(x), x = (float)(phase>>8) * 2pi/2^24 */
    if (g->nco_enable) {
        uint32_t ph = S->phase[c];
        const uint32_t step = S->nco_step[c];
        for (uint32_t n = 0; n < nb; ++n) {
            float x = (float)(ph >> 8) * ORC_NCO_K;
            lo[2 * n] = orc_cos_f32(x, ar);
            lo[2 * n + 1] = -orc_sin_f32(x, ar);
            ph += step;
        }
        S->phase[c] = ph;
        orc_cmplx_mult_cmplx_f32(iq, lo, mixed, nb, ar);
    } else {
        memcpy(mixed, iq, (size_t)nb * 2 * sizeof(float));
    }
    for (uint32_t n = 0; n < nb; ++n) { ri[n] = mixed[2 * n]; rq[n] = mixed[2 * n + 1]; }

    /* 2. decimating low-pass on both rails */
    if (g->nd_taps) {
        float *st = S->dec_state + (size_t)c * 2 * S->dec_stride;
        orc_fir_decimate_f32(S->dec_coeffs, g->nd_taps, M, st, ri, di, nb, ar);
        orc_fir_decimate_f32(S->dec_coeffs, g->nd_taps, M, st + S->dec_stride, rq, dq, nb, ar);
    } else {
        memcpy(di, ri, na * sizeof(float));
        memcpy(dq, rq, na * sizeof(float));
    }

    /* 3. demodulator */
    if (g->mode == SELENITE_MODE_AM) {
        float *z = mixed;           /* re-interleave the decimated rails */
        for (uint32_t n = 0; n < na; ++n) { z[2 * n] = di[n]; z[2 * n + 1] = dq[n]; }
        orc_cmplx_mag_f32(z, audio, na, ar);
    } else if (g->mode == SELENITE_MODE_FM) {
        /* FM (build-defined): the decimated rails run through the delay lines of the FIR pair --
         * the state update of arm_fir_f32 (arm_fir_f32.c:573-577 new samples behind the history, :947-978 copy-back),
         * taps not evaluated -- and the discriminator takes z[n] * conj(z[n-1]) (arm_cmplx_conj_f32 +
         * arm_cmplx_mult_cmplx_f32), its angle (fm_atan.h) in half turns (arm_scale_f32) */
        const uint32_t H = g->nh_taps - 1u;
        float *stI = S->fir_state + (size_t)c * 2 * S->fir_stride, *stQ = stI + S->fir_stride;
        float *z = mixed, *zc = lo, *w = ri;    /* ri, rq are contiguous: [2 * nb] */
        memcpy(stI + H, di, na * sizeof(float));
        memcpy(stQ + H, dq, na * sizeof(float));
        for (uint32_t n = 0; n < na; ++n) {
            z[2 * n] = stI[H + n];          z[2 * n + 1] = stQ[H + n];
            zc[2 * n] = stI[H + n - 1u];    zc[2 * n + 1] = stQ[H + n - 1u];
        }
        orc_cmplx_conj_f32(zc, zc, na);
        orc_cmplx_mult_cmplx_f32(z, zc, w, na, ar);
        for (uint32_t n = 0; n < na; ++n) audio[n] = orc_fm_atan2_f32(w[2 * n + 1], w[2 * n]);
        orc_scale_f32(audio, FM_AUDIO_SCALE, audio, na);
        memmove(stI, stI + na, (size_t)H * sizeof(float));
        memmove(stQ, stQ + na, (size_t)H * sizeof(float));
    } else if (g->nh_taps) {
        float *st = S->fir_state + (size_t)c * 2 * S->fir_stride;
        orc_fir_f32(S->delay_coeffs, g->nh_taps, st, di, ri, na, ar);                 /* I' */
        orc_fir_f32(S->hilb_coeffs, g->nh_taps, st + S->fir_stride, dq, rq, na, ar);  /* Q' */
        if (mode_is_upper(g->mode)) orc_sub_f32(ri, rq, audio, na);
        else orc_add_f32(ri, rq, audio, na);
    } else {
        memcpy(audio, di, na * sizeof(float));
    }

    /* 4. CW narrow filter */
    if (mode_is_cw(g->mode) && g->n_biquad) {
        float *st = S->biq_state + (size_t)c * 4 * g->n_biquad;
        orc_biquad_cascade_df1_f32(S->biquad_coeffs, g->n_biquad, st, audio, audio, na, ar);
    }

    /* 5. envelope for the AGC */
    orc_abs_f32(audio, ri, na);
    orc_max_f32(ri, na, env, NULL);
}

typedef struct {
    orc_rx *S;
    const float *iq;
    float *audio;
    uint32_t block_size, c0, c1;
} job_t;

/* per-channel AGC: whole call for channels [c0,c1) */
static void *run_channels(void *arg)
{
    job_t *j = (job_t *)arg;
    orc_rx *S = j->S;
    const selenite_rx_config *g = &S->cfg;
    const uint32_t nb = g->block, na = nb / g->decim, nblk = j->block_size / nb;
    const size_t in_stride = (size_t)j->block_size * 2, out_stride = j->block_size / g->decim;
    float *work = (float *)malloc((size_t)8 * nb * sizeof(float));
    for (uint32_t c = j->c0; c < j->c1; ++c) {
        for (uint32_t b = 0; b < nblk; ++b) {
            float *a = j->audio + c * out_stride + (size_t)b * na;
            float env;
            chain_block(S, c, j->iq + c * in_stride + (size_t)b * nb * 2, a, work, &env);
            if (g->agc_enable) {
                S->gain[c] = orc_agc_update(g, S->gain[c], env, (int)g->arith);
                orc_scale_f32(a, S->gain[c], a, na);
            }
        }
    }
    free(work);
    return NULL;
}

void orc_rx_process_f32_env(orc_rx *S, const float *iq, float *audio, uint32_t block_size,
                            const float *env_override, float *env_out)
{
    const selenite_rx_config *g = &S->cfg;
    const uint32_t nb = g->block, na = nb / g->decim, nblk = block_size / nb, C = g->channels;
    const size_t in_stride = (size_t)block_size * 2, out_stride = block_size / g->decim;
    float *work = (float *)malloc((size_t)8 * nb * sizeof(float));
    for (uint32_t b = 0; b < nblk; ++b) {
        float genv = 0.0f;
        for (uint32_t c = 0; c < C; ++c) {
            float env;
            chain_block(S, c, iq + c * in_stride + (size_t)b * nb * 2,
                        audio + c * out_stride + (size_t)b * na, work, &env);
            if (c == 0 || genv < env) genv = env;
        }
        if (env_out) env_out[b] = genv;
        if (env_override) genv = env_override[b];
        if (g->agc_enable) {
            for (uint32_t c = 0; c < C; ++c) {
                float *a = audio + c * out_stride + (size_t)b * na;
                S->gain[c] = orc_agc_update(g, S->gain[c], genv, (int)g->arith);
                orc_scale_f32(a, S->gain[c], a, na);
            }
        }
    }
    free(work);
}

void orc_rx_process_f32(orc_rx *S, const float *iq, float *audio, uint32_t block_size, int nthreads)
{
    const selenite_rx_config *g = &S->cfg;
    if (block_size == 0 || block_size % g->block != 0) return;
    if (g->agc_enable && g->agc_global) {
        orc_rx_process_f32_env(S, iq, audio, block_size, NULL, NULL);
        return;
    }
    const uint32_t C = g->channels;
    if (nthreads <= 1 || C < 2) {
        job_t j = { S, iq, audio, block_size, 0, C };
        run_channels(&j);
        return;
    }
    if ((uint32_t)nthreads > C) nthreads = (int)C;
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    job_t *jobs = (job_t *)malloc((size_t)nthreads * sizeof(job_t));
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){ S, iq, audio, block_size,
                           (uint32_t)((uint64_t)C * t / nthreads),
                           (uint32_t)((uint64_t)C * (t + 1) / nthreads) };
        pthread_create(&th[t], NULL, run_channels, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* q15 slot format: arm_q15_to_float in front, arm_float_to_q15 behind (dsp_if.c:50-67 slot) */
void orc_rx_process_q15(orc_rx *S, const int16_t *iq, int16_t *audio, uint32_t block_size, int nthreads)
{
    const selenite_rx_config *g = &S->cfg;
    const size_t nin = (size_t)g->channels * block_size * 2, nout = (size_t)g->channels * block_size / g->decim;
    float *fi = (float *)malloc(nin * sizeof(float)), *fo = (float *)malloc(nout * sizeof(float));
    orc_q15_to_float(iq, fi, (uint32_t)nin);
    orc_rx_process_f32(S, fi, fo, block_size, nthreads);
    float_to_q15(fo, audio, (uint32_t)nout, g->q15_rounding != 0);
    free(fi); free(fo);
}

int orc_rx_get_state(orc_rx *S, const selenite_rx_state_view *v)
{
    const selenite_rx_config *g = &S->cfg;
    const uint32_t C = g->channels;
    for (uint32_t c = 0; c < C; ++c) {
        if (v->dec_state && g->nd_taps > 1)
            for (int r = 0; r < 2; ++r)
                memcpy(v->dec_state + ((size_t)c * 2 + r) * (g->nd_taps - 1),
                       S->dec_state + ((size_t)c * 2 + r) * S->dec_stride, (g->nd_taps - 1) * sizeof(float));
        if (v->fir_state && g->nh_taps > 1)
            for (int r = 0; r < 2; ++r)
                memcpy(v->fir_state + ((size_t)c * 2 + r) * (g->nh_taps - 1),
                       S->fir_state + ((size_t)c * 2 + r) * S->fir_stride, (g->nh_taps - 1) * sizeof(float));
    }
    if (v->biq_state && g->n_biquad) memcpy(v->biq_state, S->biq_state, (size_t)C * 4 * g->n_biquad * sizeof(float));
    if (v->agc_gain) memcpy(v->agc_gain, S->gain, C * sizeof(float));
    if (v->nco_phase) memcpy(v->nco_phase, S->phase, C * sizeof(uint32_t));
    return SELENITE_RX_SUCCESS;
}

int orc_rx_set_state(orc_rx *S, const selenite_rx_state_view *v)
{
    const selenite_rx_config *g = &S->cfg;
    const uint32_t C = g->channels;
    for (uint32_t c = 0; c < C; ++c) {
        if (v->dec_state && g->nd_taps > 1)
            for (int r = 0; r < 2; ++r)
                memcpy(S->dec_state + ((size_t)c * 2 + r) * S->dec_stride,
                       v->dec_state + ((size_t)c * 2 + r) * (g->nd_taps - 1), (g->nd_taps - 1) * sizeof(float));
        if (v->fir_state && g->nh_taps > 1)
            for (int r = 0; r < 2; ++r)
                memcpy(S->fir_state + ((size_t)c * 2 + r) * S->fir_stride,
                       v->fir_state + ((size_t)c * 2 + r) * (g->nh_taps - 1), (g->nh_taps - 1) * sizeof(float));
    }
    if (v->biq_state && g->n_biquad) memcpy(S->biq_state, v->biq_state, (size_t)C * 4 * g->n_biquad * sizeof(float));
    if (v->agc_gain) memcpy(S->gain, v->agc_gain, C * sizeof(float));
    if (v->nco_phase) memcpy(S->phase, v->nco_phase, C * sizeof(uint32_t));
    return SELENITE_RX_SUCCESS;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic I/Q (build-defined): per channel three complex tones + uniform noise.
 * Integer phase accumulators, table-lerp sin/cos (CMSIS arithmetic), fixed f32 operation order:
 * identical bits on host and device.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_iq(float *iq, uint32_t first_channel, uint32_t nch,
                  uint64_t first_sample, uint32_t nsamp, uint64_t seed)
{
    static const float amp[3] = { 0.4f, 0.2f, 0.1f };
    for (uint32_t ci = 0; ci < nch; ++ci) {
        const uint32_t c = first_channel + ci;
        uint64_t s = seed ^ ((uint64_t)c * 0xD1B54A32D192ED03ull);
        uint32_t step[3], ph0[3];
        uint64_t r0 = splitmix64(&s), r1 = splitmix64(&s), r2 = splitmix64(&s), r3 = splitmix64(&s);
        step[0] = 0x02000000u + ((uint32_t)r0 & 0x00FFFFFFu);   /* fs/128 .. 1.5 fs/128, USB side */
        step[1] = (uint32_t)(r0 >> 32);
        step[2] = (uint32_t)r1;
        ph0[0] = (uint32_t)(r1 >> 32);
        ph0[1] = (uint32_t)r2;
        ph0[2] = (uint32_t)(r2 >> 32);
        const uint64_t nseed = r3;
        float *out = iq + (size_t)ci * nsamp * 2;
        for (uint32_t k = 0; k < nsamp; ++k) {
            const uint64_t n = first_sample + k;
            float vi = 0.0f, vq = 0.0f;
            for (int t = 0; t < 3; ++t) {
                uint32_t ph = ph0[t] + (uint32_t)n * step[t];
                float x = (float)(ph >> 8) * ORC_NCO_K;
                float cs = orc_cos_f32(x, SELENITE_ARITH_CMSIS);
                float sn = orc_sin_f32(x, SELENITE_ARITH_CMSIS);
                float pc = amp[t] * cs, ps = amp[t] * sn;
                vi = vi + pc;
                vq = vq + ps;
            }
            uint64_t h = mix64(nseed + n * 0x9E3779B97F4A7C15ull);
            float ui = (float)(uint32_t)(h >> 40) * 0x1p-24f;          /* [0,1) */
            float uq = (float)(uint32_t)((h >> 16) & 0xFFFFFFu) * 0x1p-24f;
            float ni = (ui - 0.5f) * 0.1f, nq = (uq - 0.5f) * 0.1f;
            out[2 * k] = vi + ni;
            out[2 * k + 1] = vq + nq;
        }
    }
}
