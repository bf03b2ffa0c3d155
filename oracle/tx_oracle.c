/* tx_oracle.c -- TEST ORACLE ONLY (never linked into the product): CPU restatement of the TX
 * chain of include/selenite_tx.h.  The chain is [build-defined] (the reference has no modulator,
 * SURVEY.md section 0); every step restates one CMSIS-DSP 1.5.3 primitive and is pinned bit-exactly
 * against the real function compiled from /root/reference (oracle/ref_tx.c, tests/test_tx_oracle.py).
 * Part of librx_oracle.so; uses rx_oracle.c's primitives for the steps the RX chain already has. */
#include <stdlib.h>
#include <string.h>

#include "../include/selenite_tx.h"
#include "rx_oracle.h"
#include "tx_oracle.h"

/* FilteringFunctions/arm_fir_interpolate_f32.c:136-563.  state = [P-1 history | block new samples],
 * P = numTaps / L.  Output phase j-1 (j = 1..L) of input sample n:
 *     y = sum_{t=0}^{P-1} state[n + t] * pCoeffs[(L - j) + t*L]
 * one accumulator from 0.0f, t ascending, product then add (:389-440 and the 4-sample unrolled
 * block :166-330 -- same per-output order); tail copy :445-475. */
void orc_fir_interpolate_f32(const float *coeffs, uint32_t num_taps, uint32_t L, float *state,
                             const float *src, float *dst, uint32_t block, int arith)
{
    const uint32_t P = num_taps / L;
    memcpy(state + (P - 1), src, block * sizeof(float));
    for (uint32_t n = 0; n < block; ++n)
        for (uint32_t j = 1; j <= L; ++j) {
            float acc = 0.0f;
            const float *c = coeffs + (L - j);
            for (uint32_t t = 0; t < P; ++t) {
                if (arith) acc = __builtin_fmaf(state[n + t], c[t * L], acc);
                else { float p = state[n + t] * c[t * L]; acc = acc + p; }
            }
            dst[n * L + (j - 1)] = acc;
        }
    memmove(state, state + block, (P - 1) * sizeof(float));
}

void orc_negate_f32(const float *src, float *dst, uint32_t n)            /* arm_negate_f32.c:125 */
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = -src[i];
}

void orc_offset_f32(const float *src, float offset, float *dst, uint32_t n)   /* arm_offset_f32.c:145 */
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = src[i] + offset;
}

static int upper(uint8_t m) { return m == SELENITE_MODE_USB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_CW; }
static int mode_ok(uint8_t m)
{
    return m == SELENITE_MODE_LSB || m == SELENITE_MODE_USB || m == SELENITE_MODE_CW || m == SELENITE_MODE_CWR ||
           m == SELENITE_MODE_AM || m == SELENITE_MODE_DIG || m == SELENITE_MODE_PKT;
}

struct orc_tx {
    selenite_tx_config cfg;
    float *ic, *hc, *dc;
    uint32_t *step, *phase;
    float *fir_state;        /* [C][2][nh-1 + block] */
    float *int_state;        /* [C][2][P-1 + block]  */
    float *gain;
    uint32_t fir_stride, int_stride;
};

static float *dupf(const float *p, size_t n)
{
    if (!p || !n) return NULL;
    float *q = malloc(n * sizeof(float));
    memcpy(q, p, n * sizeof(float));
    return q;
}

int orc_tx_create(orc_tx **out, const selenite_tx_config *g)
{
    *out = NULL;
    if (!g || (g->struct_size != sizeof(*g) && g->struct_size != SELENITE_TX_CONFIG_SIZE_V1) || (g->struct_size == sizeof(*g) && g->q15_rounding > 1u) || !g->channels || !g->block || !g->interp || !mode_ok(g->mode) || g->arith > 2)
        return SELENITE_RX_ARGUMENT_ERROR;
    if ((g->interp > 1) != (g->ni_taps > 0)) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->ni_taps && !g->interp_coeffs) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->nh_taps && (!g->hilb_coeffs || !g->delay_coeffs)) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->ni_taps % g->interp) return SELENITE_RX_LENGTH_ERROR;       /* arm_fir_interpolate_init_f32.c:91-96 */
    orc_tx *S = calloc(1, sizeof *S);
    S->cfg = *g;
    const uint32_t C = g->channels, P = g->ni_taps ? g->ni_taps / g->interp : 1;
    S->ic = dupf(g->interp_coeffs, g->ni_taps);
    S->hc = dupf(g->hilb_coeffs, g->nh_taps);
    S->dc = dupf(g->delay_coeffs, g->nh_taps);
    S->fir_stride = (g->nh_taps ? g->nh_taps - 1 : 0) + g->block;
    S->int_stride = (P - 1) + g->block;
    S->fir_state = calloc((size_t)C * 2 * S->fir_stride, sizeof(float));
    S->int_state = calloc((size_t)C * 2 * S->int_stride, sizeof(float));
    S->gain = malloc(C * sizeof(float));
    S->step = malloc(C * sizeof(uint32_t));
    S->phase = calloc(C, sizeof(uint32_t));
    for (uint32_t c = 0; c < C; ++c) {
        S->gain[c] = g->alc_gain_init;
        S->step[c] = g->nco_step ? g->nco_step[c] : g->nco_step_all;
    }
    *out = S;
    return SELENITE_RX_SUCCESS;
}

void orc_tx_destroy(orc_tx *S)
{
    if (!S) return;
    free(S->ic); free(S->hc); free(S->dc); free(S->fir_state); free(S->int_state);
    free(S->gain); free(S->step); free(S->phase); free(S);
}

int orc_tx_set_mode(orc_tx *S, uint8_t mode)
{
    if (!mode_ok(mode)) return SELENITE_RX_ARGUMENT_ERROR;
    S->cfg.mode = mode;
    return SELENITE_RX_SUCCESS;
}

/* the ALC gain law is the RX AGC law (rx_oracle.c orc_agc_update) on the TX parameter names */
static float alc_update(const selenite_tx_config *g, float gain, float env)
{
    selenite_rx_config r;
    memset(&r, 0, sizeof r);
    r.agc_target = g->alc_target; r.agc_attack = g->alc_attack; r.agc_decay = g->alc_decay;
    r.agc_gain_min = g->alc_gain_min; r.agc_gain_max = g->alc_gain_max; r.agc_env_floor = g->alc_env_floor;
    return orc_agc_update(&r, gain, env, 0);
}

/* one ALC block of one channel: audio[nb] -> iq[nb*L][2] */
static void tx_block(orc_tx *S, uint32_t c, const float *audio, float *iq, float *w)
{
    const selenite_tx_config *g = &S->cfg;
    const uint32_t nb = g->block, L = g->interp, no = nb * L;
    const int ar = g->arith != 0;
    float *a = w, *ri = w + nb, *rq = w + 2 * nb, *ui = w + 3 * nb, *uq = ui + no, *lo = uq + no, *z = lo + 2 * no;
    /* 1. ALC */
    memcpy(a, audio, nb * sizeof(float));
    if (g->alc_enable) {
        float env; uint32_t idx;
        orc_abs_f32(a, ri, nb);
        orc_max_f32(ri, nb, &env, &idx);
        S->gain[c] = alc_update(g, S->gain[c], env);
        orc_scale_f32(a, S->gain[c], a, nb);
    }
    /* 2. Hilbert pair */
    if (g->nh_taps) {
        float *st = S->fir_state + (size_t)c * 2 * S->fir_stride;
        orc_fir_f32(S->dc, g->nh_taps, st, a, ri, nb, ar);
        orc_fir_f32(S->hc, g->nh_taps, st + S->fir_stride, a, rq, nb, ar);
    } else {
        memcpy(ri, a, nb * sizeof(float));
        memset(rq, 0, nb * sizeof(float));
    }
    /* 3. sideband select */
    if (g->mode == SELENITE_MODE_AM) {
        orc_scale_f32(ri, 0.5f, ri, nb);
        orc_offset_f32(ri, 0.5f, ri, nb);
        memset(rq, 0, nb * sizeof(float));
    } else if (!upper(g->mode)) {
        orc_negate_f32(rq, rq, nb);
    }
    /* 4. interpolator on both rails */
    if (g->ni_taps) {
        float *st = S->int_state + (size_t)c * 2 * S->int_stride;
        orc_fir_interpolate_f32(S->ic, g->ni_taps, L, st, ri, ui, nb, ar);
        orc_fir_interpolate_f32(S->ic, g->ni_taps, L, st + S->int_stride, rq, uq, nb, ar);
    } else {
        memcpy(ui, ri, nb * sizeof(float));
        memcpy(uq, rq, nb * sizeof(float));
    }
    /* 5. NCO up-mix: LO = (cos x, +sin x); the phase rule is the RX chain's */
    for (uint32_t n = 0; n < no; ++n) { z[2 * n] = ui[n]; z[2 * n + 1] = uq[n]; }
    if (g->nco_enable) {
        uint32_t ph = S->phase[c];
        for (uint32_t n = 0; n < no; ++n) {
            const float x = (float)(ph >> 8) * 0x1.921fb6p-22f;
            lo[2 * n] = orc_cos_f32(x, 0);
            lo[2 * n + 1] = orc_sin_f32(x, 0);
            ph += S->step[c];
        }
        S->phase[c] = ph;
        orc_cmplx_mult_cmplx_f32(z, lo, iq, no, 0);
    } else {
        memcpy(iq, z, (size_t)2 * no * sizeof(float));
    }
}

void orc_tx_process_f32(orc_tx *S, const float *audio, float *iq, uint32_t block_size)
{
    const selenite_tx_config *g = &S->cfg;
    const uint32_t nb = g->block, L = g->interp, nblk = block_size / nb;
    float *w = malloc(((size_t)3 * nb + (size_t)6 * nb * L) * sizeof(float));
    for (uint32_t c = 0; c < g->channels; ++c)
        for (uint32_t b = 0; b < nblk; ++b)
            tx_block(S, c, audio + (size_t)c * block_size + (size_t)b * nb,
                     iq + ((size_t)c * block_size + (size_t)b * nb) * L * 2, w);
    free(w);
}

void orc_tx_process_q15(orc_tx *S, const int16_t *audio, int16_t *iq, uint32_t block_size)
{
    const size_t ni = (size_t)S->cfg.channels * block_size, no = ni * S->cfg.interp * 2;
    float *fi = malloc(ni * sizeof(float)), *fo = malloc(no * sizeof(float));
    orc_q15_to_float(audio, fi, (uint32_t)ni);
    orc_tx_process_f32(S, fi, fo, block_size);
    if (S->cfg.q15_rounding) orc_float_to_q15_rounding(fo, iq, (uint32_t)no);    /* the ARM_MATH_ROUNDING build of the same function (rx_oracle.c) */
    else orc_float_to_q15(fo, iq, (uint32_t)no);
    free(fi); free(fo);
}

int orc_tx_get_state(orc_tx *S, const selenite_tx_state_view *v)
{
    const selenite_tx_config *g = &S->cfg;
    const uint32_t C = g->channels, nh1 = g->nh_taps ? g->nh_taps - 1 : 0, p1 = g->ni_taps ? g->ni_taps / g->interp - 1 : 0;
    for (uint32_t c = 0; c < C; ++c)
        for (int r = 0; r < 2; ++r) {
            if (v->fir_state && nh1) memcpy(v->fir_state + ((size_t)c * 2 + r) * nh1, S->fir_state + ((size_t)c * 2 + r) * S->fir_stride, nh1 * sizeof(float));
            if (v->interp_state && p1) memcpy(v->interp_state + ((size_t)c * 2 + r) * p1, S->int_state + ((size_t)c * 2 + r) * S->int_stride, p1 * sizeof(float));
        }
    if (v->alc_gain) memcpy(v->alc_gain, S->gain, C * sizeof(float));
    if (v->nco_phase) memcpy(v->nco_phase, S->phase, C * sizeof(uint32_t));
    return 0;
}
