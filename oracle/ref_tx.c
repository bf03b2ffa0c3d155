/*
 * ref_tx.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The build-defined TX chain of include/selenite_tx.h composed of the REAL CMSIS-DSP 1.5.3 functions
 * (compiled from the sources under /root/reference by oracle/Makefile into oracle/_ref/libcmsis_ref.so;
 * nothing is copied into the repo).  Pins oracle/tx_oracle.c bit for bit (tests/test_tx_oracle.py)
 * and generates the committed TX fixtures (tests/golden/make_tx_golden.py).  Every step is literally
 * "CMSIS-DSP function X with arguments Y".
 */
#include "arm_math.h"
#include "../include/selenite_tx.h"

#include <stdlib.h>
#include <string.h>

void arm_float_to_q15_rounding(float32_t *pSrc, q15_t *pDst, uint32_t blockSize);   /* arm_float_to_q15.c built with -DARM_MATH_ROUNDING under this name (oracle/Makefile) */

#define REF_NCO_K 0x1.921fb6p-22f

typedef struct ref_tx {
    selenite_tx_config cfg;
    float *ic, *hc, *dc;
    uint32_t *step, *phase;
    arm_fir_instance_f32 *fir;                    /* [C][2]  0 = delay, 1 = Hilbert */
    arm_fir_interpolate_instance_f32 *itp;        /* [C][2]  I, Q */
    float *fir_state, *int_state, *gain;
    size_t fir_stride, int_stride;
} ref_tx;

static float *dupf(const float *p, size_t n)
{
    if (!p || !n) return NULL;
    float *q = (float *)malloc(n * sizeof(float));
    memcpy(q, p, n * sizeof(float));
    return q;
}
static int upper(uint8_t m) { return m == SELENITE_MODE_USB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_CW; }

int ref_tx_create(ref_tx **out, const selenite_tx_config *g)
{
    *out = NULL;
    if (!g || !g->channels || !g->block || !g->interp || g->arith != SELENITE_ARITH_CMSIS) return ARM_MATH_ARGUMENT_ERROR;
    ref_tx *S = (ref_tx *)calloc(1, sizeof *S);
    S->cfg = *g;
    const uint32_t C = g->channels, P = g->ni_taps ? g->ni_taps / g->interp : 1;
    S->ic = dupf(g->interp_coeffs, g->ni_taps);
    S->hc = dupf(g->hilb_coeffs, g->nh_taps);
    S->dc = dupf(g->delay_coeffs, g->nh_taps);
    S->fir_stride = (g->nh_taps ? g->nh_taps - 1 : 0) + g->block;
    S->int_stride = (P - 1) + g->block;
    S->fir_state = (float *)calloc((size_t)C * 2 * S->fir_stride + 1, sizeof(float));
    S->int_state = (float *)calloc((size_t)C * 2 * S->int_stride + 1, sizeof(float));
    S->fir = (arm_fir_instance_f32 *)calloc((size_t)C * 2, sizeof *S->fir);
    S->itp = (arm_fir_interpolate_instance_f32 *)calloc((size_t)C * 2, sizeof *S->itp);
    S->gain = (float *)malloc(C * sizeof(float));
    S->step = (uint32_t *)malloc(C * sizeof(uint32_t));
    S->phase = (uint32_t *)calloc(C, sizeof(uint32_t));
    for (uint32_t c = 0; c < C; ++c) {
        S->gain[c] = g->alc_gain_init;
        S->step[c] = g->nco_step ? g->nco_step[c] : g->nco_step_all;
        for (int r = 0; r < 2; ++r) {
            if (g->nh_taps)
                arm_fir_init_f32(&S->fir[2 * c + r], (uint16_t)g->nh_taps, r ? S->hc : S->dc,
                                 S->fir_state + ((size_t)2 * c + r) * S->fir_stride, g->block);
            if (g->ni_taps) {
                arm_status st = arm_fir_interpolate_init_f32(&S->itp[2 * c + r], (uint8_t)g->interp, (uint16_t)g->ni_taps,
                                                             S->ic, S->int_state + ((size_t)2 * c + r) * S->int_stride, g->block);
                if (st != ARM_MATH_SUCCESS) return st;        /* LENGTH_ERROR path (harness object leaks: test only) */
            }
        }
    }
    *out = S;
    return ARM_MATH_SUCCESS;
}

void ref_tx_destroy(ref_tx *S)
{
    if (!S) return;
    free(S->ic); free(S->hc); free(S->dc); free(S->fir_state); free(S->int_state); free(S->fir); free(S->itp);
    free(S->gain); free(S->step); free(S->phase); free(S);
}

int ref_tx_set_mode(ref_tx *S, uint8_t mode) { S->cfg.mode = mode; return ARM_MATH_SUCCESS; }

static float alc_update(const selenite_tx_config *g, float gain, float env)     /* the RX harness' AGC law */
{
    float e = (env < g->alc_env_floor) ? g->alc_env_floor : env;
    float d = g->alc_target / e;
    if (d > g->alc_gain_max) d = g->alc_gain_max;
    if (d < g->alc_gain_min) d = g->alc_gain_min;
    float diff = d - gain;
    float rate = (diff < 0.0f) ? g->alc_attack : g->alc_decay;
    float p = rate * diff;
    return gain + p;
}

void ref_tx_process_f32(ref_tx *S, const float *audio, float *iq, uint32_t block_size)
{
    const selenite_tx_config *g = &S->cfg;
    const uint32_t nb = g->block, L = g->interp, no = nb * L, nblk = block_size / nb;
    float *w = (float *)malloc(((size_t)3 * nb + (size_t)6 * no) * sizeof(float));
    float *a = w, *ri = w + nb, *rq = w + 2 * nb, *ui = w + 3 * nb, *uq = ui + no, *lo = uq + no, *z = lo + 2 * no;
    for (uint32_t c = 0; c < g->channels; ++c)
        for (uint32_t b = 0; b < nblk; ++b) {
            float *out = iq + ((size_t)c * block_size + (size_t)b * nb) * L * 2;
            memcpy(a, audio + (size_t)c * block_size + (size_t)b * nb, nb * sizeof(float));
            if (g->alc_enable) {
                float env; uint32_t idx;
                arm_abs_f32(a, ri, nb);
                arm_max_f32(ri, nb, &env, &idx);
                S->gain[c] = alc_update(g, S->gain[c], env);
                arm_scale_f32(a, S->gain[c], a, nb);
            }
            if (g->nh_taps) {
                arm_fir_f32(&S->fir[2 * c], a, ri, nb);
                arm_fir_f32(&S->fir[2 * c + 1], a, rq, nb);
            } else {
                memcpy(ri, a, nb * sizeof(float));
                memset(rq, 0, nb * sizeof(float));
            }
            if (g->mode == SELENITE_MODE_AM) {
                arm_scale_f32(ri, 0.5f, ri, nb);
                arm_offset_f32(ri, 0.5f, ri, nb);
                memset(rq, 0, nb * sizeof(float));
            } else if (!upper(g->mode)) {
                arm_negate_f32(rq, rq, nb);
            }
            if (g->ni_taps) {
                arm_fir_interpolate_f32(&S->itp[2 * c], ri, ui, nb);
                arm_fir_interpolate_f32(&S->itp[2 * c + 1], rq, uq, nb);
            } else {
                memcpy(ui, ri, nb * sizeof(float));
                memcpy(uq, rq, nb * sizeof(float));
            }
            for (uint32_t n = 0; n < no; ++n) { z[2 * n] = ui[n]; z[2 * n + 1] = uq[n]; }
            if (g->nco_enable) {
                uint32_t ph = S->phase[c];
                for (uint32_t n = 0; n < no; ++n) {
                    const float x = (float)(ph >> 8) * REF_NCO_K;
                    lo[2 * n] = arm_cos_f32(x);
                    lo[2 * n + 1] = arm_sin_f32(x);
                    ph += S->step[c];
                }
                S->phase[c] = ph;
                arm_cmplx_mult_cmplx_f32(z, lo, out, no);
            } else {
                memcpy(out, z, (size_t)2 * no * sizeof(float));
            }
        }
    free(w);
}

void ref_tx_process_q15(ref_tx *S, const int16_t *audio, int16_t *iq, uint32_t block_size)
{
    const size_t ni = (size_t)S->cfg.channels * block_size, no = ni * S->cfg.interp * 2;
    float *fi = (float *)malloc(ni * sizeof(float)), *fo = (float *)malloc(no * sizeof(float));
    arm_q15_to_float((q15_t *)audio, fi, (uint32_t)ni);
    ref_tx_process_f32(S, fi, fo, block_size);
    if (S->cfg.q15_rounding) arm_float_to_q15_rounding(fo, (q15_t *)iq, (uint32_t)no);   /* SupportFunctions/arm_float_to_q15.c built with -DARM_MATH_ROUNDING (oracle/Makefile) */
    else arm_float_to_q15(fo, (q15_t *)iq, (uint32_t)no);
    free(fi); free(fo);
}

int ref_tx_get_state(ref_tx *S, const selenite_tx_state_view *v)
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

/* primitive wrappers for the per-function pin tests */
void ref_fir_interpolate(const float *coeffs, uint32_t num_taps, uint32_t L, float *state, const float *src, float *dst, uint32_t block)
{
    arm_fir_interpolate_instance_f32 S = { (uint8_t)L, (uint16_t)(num_taps / L), (float *)coeffs, state };
    arm_fir_interpolate_f32(&S, (float *)src, dst, block);
}
int ref_fir_interpolate_init_status(uint32_t num_taps, uint32_t L, uint32_t block)
{
    arm_fir_interpolate_instance_f32 S;
    float *co = (float *)calloc(num_taps + 1, sizeof(float)), *st = (float *)calloc(num_taps + block + 1, sizeof(float));
    int rc = arm_fir_interpolate_init_f32(&S, (uint8_t)L, (uint16_t)num_taps, co, st, block);
    free(co); free(st);
    return rc;
}
void ref_negate(const float *src, float *dst, uint32_t n) { arm_negate_f32((float *)src, dst, n); }
void ref_offset(const float *src, float off, float *dst, uint32_t n) { arm_offset_f32((float *)src, off, dst, n); }
