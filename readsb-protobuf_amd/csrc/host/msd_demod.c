/*
 * msd_demod.c -- the demodulator entry points with the reference's own shape (demod_2400.h:37-38):
 *
 *     void demodulate2400(struct mag_buf *mag);
 *     void demodulate2400AC(struct mag_buf *mag);
 *
 * The reference's functions find their state in the global `Modes` and hand every message to
 * useModesMessage(); here a receiver is bound once (msd_demod_bind: context + message sink) and the two
 * functions take nothing but the buffer, so the consumer loop of readsb.c:820-855 keeps its two calls:
 *
 *     demodulate2400(buf);                       ->  msd_demodulate2400(buf);
 *     if (Modes.mode_ac) demodulate2400AC(buf);  ->  if (Modes.mode_ac) msd_demodulate2400AC(buf);
 *
 * One GPU call (msd_demodulate_magbuf) does the work of both for a buffer -- the Mode A/C pass shares the
 * upload and the buffer's noise level with the Mode S pass.  msd_demodulate2400() runs it, delivers the
 * Mode S messages and keeps the buffer's Mode A/C replies; msd_demodulate2400AC() on the same buffer delivers
 * those, so the sink sees exactly the reference's order (all Mode S messages of a buffer, then its replies,
 * readsb.c:826-829).  Host C: a few pointer moves per message.
 */
#include "modes_hip_readsb.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static struct {
    msd_ctx *ctx;
    msd_message_fn sink;
    void *user;
    int mode_ac;
    /* the Mode A/C replies of the buffer msd_demodulate2400 saw last */
    const struct msd_mag_buf *held_for;
    msd_message *held;
    size_t nheld, cap;
    char err[256];
} D;

int msd_demod_bind(msd_ctx *ctx, int mode_ac, msd_message_fn sink, void *user)
{
    D.ctx = ctx;
    D.sink = sink;
    D.user = user;
    D.mode_ac = mode_ac;
    D.held_for = NULL;
    D.nheld = 0;
    D.err[0] = 0;
    if (!ctx) { /* unbind: drop the storage as well */
        free(D.held);
        D.held = NULL;
        D.cap = 0;
    }
    return 0;
}

const char *msd_demod_error(void)
{
    return D.err;
}

static void split_sink(const msd_message *mm, void *user)
{
    (void)user;
    if (mm->msgtype != 32) { /* Mode S: straight through (useModesMessage, demod_2400.c:404) */
        if (D.sink)
            D.sink(mm, D.user);
        return;
    }
    /* Mode A/C (mode_ac.c:171): the reference produces these in its second call */
    if (D.nheld == D.cap) {
        const size_t cap = D.cap ? 2 * D.cap : 256;
        msd_message *p = realloc(D.held, cap * sizeof *p);
        if (!p) {
            snprintf(D.err, sizeof D.err, "out of memory holding Mode A/C replies");
            return;
        }
        D.held = p;
        D.cap = cap;
    }
    D.held[D.nheld++] = *mm;
}

void msd_demodulate2400(struct msd_mag_buf *mag)
{
    D.nheld = 0;
    D.held_for = NULL;
    D.err[0] = 0; /* msd_demod_error() speaks for the latest buffer only */
    if (!mag)
        return;
    if (!D.ctx) {
        snprintf(D.err, sizeof D.err, "msd_demodulate2400: no receiver bound (msd_demod_bind)");
        return;
    }
    const int rc = msd_demodulate_magbuf(D.ctx, mag->data, mag->validLength, mag->overlap, mag->sampleTimestamp,
                                         mag->sysTimestamp, mag->mean_level, mag->mean_power, split_sink, NULL);
    if (rc) {
        snprintf(D.err, sizeof D.err, "msd_demodulate2400: %s", msd_last_error(D.ctx));
        D.nheld = 0;
        return;
    }
    D.held_for = mag;
}

void msd_demodulate2400AC(struct msd_mag_buf *mag)
{
    if (!mag || mag != D.held_for) {
        if (mag && D.ctx && D.mode_ac)
            snprintf(D.err, sizeof D.err, "msd_demodulate2400AC: call msd_demodulate2400 on the same buffer first");
        return;
    }
    for (size_t i = 0; i < D.nheld; ++i)
        if (D.sink)
            D.sink(&D.held[i], D.user);
    D.nheld = 0;
    D.held_for = NULL;
}
