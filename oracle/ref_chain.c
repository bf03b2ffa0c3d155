/*
 * ref_chain.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Composes the REAL CMSIS-DSP 1.5.3 functions (compiled from the sources where they lie under
 * /root/reference by oracle/Makefile -> oracle/_ref/libcmsis_ref.so; nothing is copied into the
 * repo) into the build-defined RX chain of DESIGN.md.  Used only in the build container to
 *   (1) pin oracle/rx_oracle.c bit-for-bit (tests/test_oracle_vs_ref.py), and
 *   (2) generate the committed fixtures under tests/golden/ (tests/golden/make_golden.py).
 * It cannot travel to the GPU box (/root/reference does not exist there).
 *
 * Every step below is literally "CMSIS-DSP function X with arguments Y".
 */
#include "arm_math.h"
#include "arm_common_tables.h"
#include "../include/selenite_rx.h"
#include "fm_atan.h"                    /* the build-defined arctangent of the FM discriminator: CMSIS-DSP 1.5.3 has none */

#include <stdlib.h>
#include <string.h>

/* SupportFunctions/arm_float_to_q15.c compiled a second time with -DARM_MATH_ROUNDING under this name (oracle/Makefile) */
void arm_float_to_q15_rounding(float32_t *pSrc, q15_t *pDst, uint32_t blockSize);

#define REF_NCO_K 0x1.921fb6p-22f      /* 2*pi / 2^24 in f32 (same constant as the oracle) */

typedef struct ref_rx {
    selenite_rx_config cfg;
    float *dec_coeffs, *hilb_coeffs, *delay_coeffs, *biquad_coeffs;
    uint32_t *nco_step;
    /* one CMSIS instance + state buffer per channel and rail */
    arm_fir_decimate_instance_f32 *dec;   /* [C][2] */
    arm_fir_instance_f32 *fir;            /* [C][2]  0 = delay (I), 1 = Hilbert (Q) */
    arm_biquad_casd_df1_inst_f32 *biq;    /* [C] */
    float *dec_state, *fir_state, *biq_state, *gain;
    uint32_t *phase;
    size_t dec_stride, fir_stride;
} ref_rx;

static float *dupf(const float *p, size_t n)
{
    if (!p || !n) return NULL;
    float *q = (float *)malloc(n * sizeof(float));
    memcpy(q, p, n * sizeof(float));
    return q;
}

static int mode_valid(uint8_t m, uint32_t nh_taps)
{
    if (m == SELENITE_MODE_FM) return nh_taps >= 2;
    return m == SELENITE_MODE_LSB || m == SELENITE_MODE_USB || m == SELENITE_MODE_CW ||
           m == SELENITE_MODE_CWR || m == SELENITE_MODE_AM || m == SELENITE_MODE_DIG ||
           m == SELENITE_MODE_PKT;
}
static int mode_is_cw(uint8_t m) { return m == SELENITE_MODE_CW || m == SELENITE_MODE_CWR; }
static int mode_is_upper(uint8_t m)
{
    return m == SELENITE_MODE_USB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_CW;
}

int ref_rx_create(ref_rx **out, const selenite_rx_config *cfg)
{
    *out = NULL;
    if (!cfg || cfg->channels == 0 || cfg->block == 0 || cfg->decim == 0) return ARM_MATH_ARGUMENT_ERROR;
    if (!mode_valid(cfg->mode, cfg->nh_taps)) return ARM_MATH_ARGUMENT_ERROR;
    if (cfg->arith != SELENITE_ARITH_CMSIS) return ARM_MATH_ARGUMENT_ERROR;  /* CMSIS has one arithmetic */
    if (cfg->nd_taps == 0 && cfg->decim != 1) return ARM_MATH_ARGUMENT_ERROR;
    ref_rx *S = (ref_rx *)calloc(1, sizeof(*S));
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
    S->dec = (arm_fir_decimate_instance_f32 *)calloc((size_t)C * 2, sizeof(*S->dec));
    S->fir = (arm_fir_instance_f32 *)calloc((size_t)C * 2, sizeof(*S->fir));
    S->biq = (arm_biquad_casd_df1_inst_f32 *)calloc(C, sizeof(*S->biq));
    for (uint32_t c = 0; c < C; ++c) {
        S->gain[c] = cfg->agc_gain_init;
        for (int r = 0; r < 2; ++r) {
            if (cfg->nd_taps) {
                arm_status st = arm_fir_decimate_init_f32(&S->dec[2 * c + r], (uint16_t)cfg->nd_taps,
                        (uint8_t)cfg->decim, S->dec_coeffs,
                        S->dec_state + ((size_t)2 * c + r) * S->dec_stride, cfg->block);
                if (st != ARM_MATH_SUCCESS) { free(S); return st; }   /* LENGTH_ERROR path */
            }
            if (cfg->nh_taps)
                arm_fir_init_f32(&S->fir[2 * c + r], (uint16_t)cfg->nh_taps,
                                 r ? S->hilb_coeffs : S->delay_coeffs,
                                 S->fir_state + ((size_t)2 * c + r) * S->fir_stride,
                                 cfg->block / cfg->decim);
        }
        if (cfg->n_biquad)
            arm_biquad_cascade_df1_init_f32(&S->biq[c], (uint8_t)cfg->n_biquad, S->biquad_coeffs,
                                            S->biq_state + (size_t)c * 4 * cfg->n_biquad);
    }
    if (cfg->block % cfg->decim != 0) { return ARM_MATH_LENGTH_ERROR; }
    *out = S;
    return ARM_MATH_SUCCESS;
}

void ref_rx_destroy(ref_rx *S)
{
    if (!S) return;
    free(S->dec_coeffs); free(S->hilb_coeffs); free(S->delay_coeffs); free(S->biquad_coeffs);
    free(S->nco_step); free(S->dec_state); free(S->fir_state); free(S->biq_state);
    free(S->gain); free(S->phase); free(S->dec); free(S->fir); free(S->biq); free(S);
}

int ref_rx_set_mode(ref_rx *S, uint8_t mode)
{
    if (!mode_valid(mode, S->cfg.nh_taps)) return ARM_MATH_ARGUMENT_ERROR;
    S->cfg.mode = mode;
    return ARM_MATH_SUCCESS;
}

static float agc_update(const selenite_rx_config *g, float gain, float env)
{
    float e = (env < g->agc_env_floor) ? g->agc_env_floor : env;
    float d = g->agc_target / e;
    if (d > g->agc_gain_max) d = g->agc_gain_max;
    if (d < g->agc_gain_min) d = g->agc_gain_min;
    float diff = d - gain;
    float rate = (diff < 0.0f) ? g->agc_attack : g->agc_decay;
    float p = rate * diff;
    return gain + p;
}

static void chain_block(ref_rx *S, uint32_t c, const float *iq, float *audio, float *work, float *env)
{
    const selenite_rx_config *g = &S->cfg;
    const uint32_t nb = g->block, M = g->decim, na = nb / M;
    float *mixed = work, *lo = work + 2 * nb, *ri = work + 4 * nb, *rq = work + 5 * nb;
    float *di = work + 6 * nb, *dq = work + 7 * nb;
    uint32_t idx;

    if (g->nco_enable) {
        uint32_t ph = S->phase[c];
        const uint32_t step = S->nco_step[c];
        for (uint32_t n = 0; n < nb; ++n) {
            float x = (float)(ph >> 8) * REF_NCO_K;
            lo[2 * n] = arm_cos_f32(x);
            lo[2 * n + 1] = -arm_sin_f32(x);
            ph += step;
        }
        S->phase[c] = ph;
        arm_cmplx_mult_cmplx_f32((float *)iq, lo, mixed, nb);
    } else {
        memcpy(mixed, iq, (size_t)nb * 2 * sizeof(float));
    }
    for (uint32_t n = 0; n < nb; ++n) { ri[n] = mixed[2 * n]; rq[n] = mixed[2 * n + 1]; }

    if (g->nd_taps) {
        arm_fir_decimate_f32(&S->dec[2 * c], ri, di, nb);
        arm_fir_decimate_f32(&S->dec[2 * c + 1], rq, dq, nb);
    } else {
        memcpy(di, ri, na * sizeof(float));
        memcpy(dq, rq, na * sizeof(float));
    }

    if (g->mode == SELENITE_MODE_AM) {
        float *z = mixed;
        for (uint32_t n = 0; n < na; ++n) { z[2 * n] = di[n]; z[2 * n + 1] = dq[n]; }
        arm_cmplx_mag_f32(z, audio, na);
    } else if (g->mode == SELENITE_MODE_FM) {
        /* the delay lines run: arm_fir_f32 itself moves the state (its output is not used); the sample in front of the
         * block is the newest entry of the history, read before the call */
        float *z = mixed, *zc = lo, *w = ri;
        const uint32_t H = g->nh_taps - 1u;
        const float pI = S->fir[2 * c].pState[H - 1u], pQ = S->fir[2 * c + 1].pState[H - 1u];
        for (uint32_t n = 0; n < na; ++n) {
            z[2 * n] = di[n];                      z[2 * n + 1] = dq[n];
            zc[2 * n] = n ? di[n - 1u] : pI;       zc[2 * n + 1] = n ? dq[n - 1u] : pQ;
        }
        arm_fir_f32(&S->fir[2 * c], di, audio, na);
        arm_fir_f32(&S->fir[2 * c + 1], dq, audio, na);
        float *zk = di;                            /* (di, dq are contiguous and free now: [2 * nb]) */
        arm_cmplx_conj_f32(zc, zk, na);
        arm_cmplx_mult_cmplx_f32(z, zk, w, na);
        for (uint32_t n = 0; n < na; ++n) audio[n] = fm_atan2_f32(w[2 * n + 1], w[2 * n]);
        arm_scale_f32(audio, FM_AUDIO_SCALE, audio, na);
    } else if (g->nh_taps) {
        arm_fir_f32(&S->fir[2 * c], di, ri, na);
        arm_fir_f32(&S->fir[2 * c + 1], dq, rq, na);
        if (mode_is_upper(g->mode)) arm_sub_f32(ri, rq, audio, na);
        else arm_add_f32(ri, rq, audio, na);
    } else {
        memcpy(audio, di, na * sizeof(float));
    }

    if (mode_is_cw(g->mode) && g->n_biquad)
        arm_biquad_cascade_df1_f32(&S->biq[c], audio, audio, na);

    arm_abs_f32(audio, ri, na);
    arm_max_f32(ri, na, env, &idx);
}

/* env_override / env_out as in orc_rx_process_f32_env (may be NULL) */
void ref_rx_process_f32_env(ref_rx *S, const float *iq, float *audio, uint32_t block_size,
                            const float *env_override, float *env_out)
{
    const selenite_rx_config *g = &S->cfg;
    const uint32_t nb = g->block, na = nb / g->decim, nblk = block_size / nb, C = g->channels;
    const size_t in_stride = (size_t)block_size * 2, out_stride = block_size / g->decim;
    float *work = (float *)malloc((size_t)8 * nb * sizeof(float));
    const int global = g->agc_enable && g->agc_global;
    for (uint32_t b = 0; b < nblk; ++b) {
        float genv = 0.0f;
        for (uint32_t c = 0; c < C; ++c) {
            float env;
            float *a = audio + c * out_stride + (size_t)b * na;
            chain_block(S, c, iq + c * in_stride + (size_t)b * nb * 2, a, work, &env);
            if (c == 0 || genv < env) genv = env;
            if (g->agc_enable && !global) {
                S->gain[c] = agc_update(g, S->gain[c], env);
                arm_scale_f32(a, S->gain[c], a, na);
            }
        }
        if (env_out) env_out[b] = genv;
        if (env_override) genv = env_override[b];
        if (global) {
            for (uint32_t c = 0; c < C; ++c) {
                float *a = audio + c * out_stride + (size_t)b * na;
                S->gain[c] = agc_update(g, S->gain[c], genv);
                arm_scale_f32(a, S->gain[c], a, na);
            }
        }
    }
    free(work);
}

void ref_rx_process_f32(ref_rx *S, const float *iq, float *audio, uint32_t block_size)
{
    if (block_size == 0 || block_size % S->cfg.block != 0) return;
    ref_rx_process_f32_env(S, iq, audio, block_size, NULL, NULL);
}

void ref_rx_process_q15(ref_rx *S, const int16_t *iq, int16_t *audio, uint32_t block_size)
{
    const selenite_rx_config *g = &S->cfg;
    const size_t nin = (size_t)g->channels * block_size * 2, nout = (size_t)g->channels * block_size / g->decim;
    float *fi = (float *)malloc(nin * sizeof(float)), *fo = (float *)malloc(nout * sizeof(float));
    arm_q15_to_float((q15_t *)iq, fi, (uint32_t)nin);
    ref_rx_process_f32(S, fi, fo, block_size);
    if (g->q15_rounding) arm_float_to_q15_rounding(fo, audio, (uint32_t)nout);   /* the same source file built with -DARM_MATH_ROUNDING (Makefile) */
    else arm_float_to_q15(fo, audio, (uint32_t)nout);
    free(fi); free(fo);
}

int ref_rx_get_state(ref_rx *S, const selenite_rx_state_view *v)
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
    return 0;
}

/* ---- thin wrappers so Python (ctypes) can drive single primitives without building CMSIS
 * instance structs ---- */
const float *ref_sin_table(void) { return sinTable_f32; }

void ref_fir_decimate(const float *coeffs, uint32_t num_taps, uint32_t M, float *state,
                      const float *src, float *dst, uint32_t block)
{
    arm_fir_decimate_instance_f32 S;
    S.M = (uint8_t)M; S.numTaps = (uint16_t)num_taps; S.pCoeffs = (float *)coeffs; S.pState = state;
    arm_fir_decimate_f32(&S, (float *)src, dst, block);
}
int ref_fir_decimate_init_status(uint32_t num_taps, uint32_t M, uint32_t block)
{
    arm_fir_decimate_instance_f32 S;
    float *st = (float *)calloc(num_taps + block, sizeof(float));
    float c = 0.0f;
    arm_status r = arm_fir_decimate_init_f32(&S, (uint16_t)num_taps, (uint8_t)M, &c, st, block);
    free(st);
    return (int)r;
}
void ref_fir(const float *coeffs, uint32_t num_taps, float *state, const float *src, float *dst, uint32_t block)
{
    arm_fir_instance_f32 S;
    S.numTaps = (uint16_t)num_taps; S.pCoeffs = (float *)coeffs; S.pState = state;
    arm_fir_f32(&S, (float *)src, dst, block);
}
void ref_biquad(const float *coeffs, uint32_t stages, float *state, const float *src, float *dst, uint32_t block)
{
    arm_biquad_casd_df1_inst_f32 S;
    S.numStages = stages; S.pCoeffs = (float *)coeffs; S.pState = state;
    arm_biquad_cascade_df1_f32(&S, (float *)src, dst, block);
}
void ref_sin_cos(const float *x, float *s, float *c, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) { s[i] = arm_sin_f32(x[i]); c[i] = arm_cos_f32(x[i]); }
}
