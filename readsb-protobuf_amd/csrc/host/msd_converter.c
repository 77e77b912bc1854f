/*
 * msd_converter.c -- init_converter / iq_convert_fn / cleanup_converter of convert.h:27-45 over the
 * GPU converter of modes_hip.h (msd_convert): the state the factory hands out carries the GPU
 * context, the function it returns has the reference's exact signature.  Host C; see
 * include/modes_hip_readsb.h.
 */
#include "modes_hip_readsb.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct converter_state {
    msd_ctx *ctx;
    char err[200];
};

static int g_device = 0;
static int g_q11_bits = 0;

int msd_converter_set_sc16q11_table_bits(int bits)
{
    if (bits < 0 || bits > 11)
        return -EINVAL; /* (used to become 0, the float path, silently -- while the fused path refused the same value) */
    g_q11_bits = bits;
    return 0;
}

void msd_converter_set_device(int device)
{
    g_device = device;
}

msd_ctx *msd_converter_context(struct converter_state *state)
{
    return state ? state->ctx : NULL;
}

const char *msd_converter_error(const struct converter_state *state)
{
    return state ? state->err : "";
}

/* iq_convert_fn (convert.h:33-38) */
static void convert_on_gpu(void *iq_data, uint16_t *mag_data, unsigned nsamples, struct converter_state *state,
                           double *out_mean_level, double *out_mean_power)
{
    int rc = state && state->ctx ? msd_convert(state->ctx, iq_data, mag_data, nsamples, out_mean_level, out_mean_power)
                                 : -EINVAL;
    if (rc) { /* the reference's converters cannot fail: leave silence and a message */
        if (state)
            snprintf(state->err, sizeof state->err, "convert: %s",
                     state->ctx ? msd_last_error(state->ctx) : "no converter state");
        if (mag_data)
            memset(mag_data, 0, (size_t)nsamples * sizeof *mag_data);
        if (out_mean_level)
            *out_mean_level = 0;
        if (out_mean_power)
            *out_mean_power = 0;
    } else if (state) {
        state->err[0] = 0;
    }
}

msd_iq_convert_fn msd_init_converter(msd_input_format_t format, double sample_rate, int filter_dc,
                                     struct converter_state **out_state)
{
    if (!out_state)
        return NULL;
    *out_state = NULL;
    int fmt;
    switch ((int)format) {
    case 0: fmt = MSD_FMT_UC8; break;     /* INPUT_UC8 */
    case 1: fmt = MSD_FMT_SC16; break;    /* INPUT_SC16 */
    case 2: fmt = MSD_FMT_SC16Q11; break; /* INPUT_SC16Q11 */
    default: return NULL;                 /* "no suitable converter", convert.c:466-470 */
    }
    if (filter_dc && !(sample_rate >= 1.0))
        return NULL; /* dc_b = exp(-2 pi / sample_rate), convert.c:479-482, needs a rate */
    struct converter_state *st = calloc(1, sizeof *st);
    if (!st)
        return NULL;
    msd_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = g_device;
    cfg.format = fmt;
    if (filter_dc) { /* convert_*_generic: the state (z1_I, z1_Q) is the context's, carried from call to call (convert.c:28-33) */
        cfg.flags |= MSD_CFG_DC_FILTER;
        cfg.sample_rate = sample_rate; /* the DC block's constant follows the caller's rate, as init_converter's does */
    }
    cfg.sc16q11_table_bits = (fmt == MSD_FMT_SC16Q11 && !filter_dc) ? g_q11_bits : 0; /* #if defined(SC16Q11_TABLE_BITS), convert.c:437 */
    cfg.preamble_threshold = 58; /* the converter does not demodulate; msd_set_preamble_threshold etc. apply */
    cfg.nfix_crc = 1;
    cfg.max_batch_samples = MSD_CHUNK_SAMPLES; /* a block of MODES_MAG_BUF_SAMPLES, sdr_ifile.c:140 */
    if (msd_create(&cfg, &st->ctx)) {
        free(st);
        return NULL;
    }
    *out_state = st;
    return convert_on_gpu;
}

void msd_cleanup_converter(struct converter_state *state)
{
    if (!state)
        return;
    if (state->ctx)
        msd_destroy(state->ctx);
    free(state);
}
