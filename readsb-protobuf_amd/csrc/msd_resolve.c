/*
 * msd_resolve.c -- the ordered resolve stage: replays demodulate2400's sequential state machine
 * (skip-ahead, ICAO filter, ifile clock, counters) over the candidate lists the GPU produced.
 *
 * Host C on purpose: this is a few hundred nanoseconds of pointer-chasing per candidate with a
 * strict order dependence; the data-parallel work (IQ->magnitude, preamble tests, bit slicing,
 * CRC, syndrome lookup, signal power) is all done on the GPU before this runs.
 */
#include "msd_internal.h"
#include "modes_hip.h"

#include <math.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------- */
/* ICAO address filter -- icao_filter.c semantics                                           */
/* ---------------------------------------------------------------------------------------- */

#define SLOTS 8192u
#define VACANT 0xFFFFFFFFu

static uint32_t hash24(uint32_t a) /* icao_filter.c:44-65 */
{
    uint32_t h = 0;
    h += a & 0xff;         h += h << 10; h ^= h >> 6;
    h += (a >> 8) & 0xff;  h += h << 10; h ^= h >> 6;
    h += (a >> 16) & 0xff; h += h << 10; h ^= h >> 6;
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return h & (SLOTS - 1);
}

static void filter_init(msd_filter *f) /* icao_filter.c:67-71 */
{
    memset(f->slot, 0xFF, sizeof f->slot);
    f->active = 0;
    f->next_flip = 0;
}

static void filter_add(msd_filter *f, uint32_t addr) /* icao_filter.c:76-97 */
{
    uint32_t *t = f->slot[f->active];
    uint32_t start = hash24(addr), h = start;
    while (t[h] != VACANT && t[h] != addr) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            return; /* table full: the reference gives up on both inserts */
    }
    if (t[h] == VACANT)
        t[h] = addr;

    uint32_t low = addr & 0xffffu;
    start = h = hash24(low);
    while (t[h] != VACANT && (t[h] & 0xffffu) != low) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            return;
    }
    if (t[h] == VACANT)
        t[h] = addr;
}

static int filter_test(const msd_filter *f, uint32_t addr) /* icao_filter.c:99-119 */
{
    uint32_t start = hash24(addr);
    for (int w = 0; w < 2; ++w) {
        const uint32_t *t = f->slot[w];
        uint32_t h = start;
        while (t[h] != VACANT && t[h] != addr) {
            h = (h + 1) & (SLOTS - 1);
            if (h == start)
                break;
        }
        if (t[h] == addr)
            return 1;
    }
    return 0;
}

static void filter_expire(msd_filter *f, uint64_t now) /* icao_filter.c:150-164 */
{
    if (now >= f->next_flip) {
        int other = f->active ^ 1;
        memset(f->slot[other], 0xFF, sizeof f->slot[other]);
        f->active = other;
        f->next_flip = now + 60000u;
    }
}

/* ---------------------------------------------------------------------------------------- */

void msd_resolver_reset(msd_resolver *r)
{
    filter_init(&r->filter);
    r->ifile_now = 0;
    r->sample_counter = 0;
    if (r->stats)
        memset(r->stats, 0, sizeof *r->stats);
}

/* scoreModesMessage (mode_s.c:311-409) given what the GPU already derived for this try */
static int score_try(const msd_try *t, int known)
{
    int df = t->msg[0] >> 3;
    int nerr = (t->errbit != 0xff);
    switch (df) {
    case 11:
        if ((t->crc & 0x7f) == 0)
            return (known ? 1600 : 750) / (nerr + 1);
        return known ? 1000 / (nerr + 1) : -1;
    case 17: case 18:
        return (known ? 1800 : 1400) / (nerr + 1);
    case 20: case 21:
        return known ? 1000 : -2;
    default: /* 0, 4, 5, 16, 24: address/parity */
        return known ? 1000 : -1;
    }
}

/* One buffer of demodulate2400 (demod_2400.c:236-428) over hits[*hi..]; everything except the
 * signal-power bookkeeping, which msd_resolve_power() adds once the sums are known. */
static void resolve_mode_s(msd_resolver *r, uint64_t batch_chunk0, uint32_t b, uint32_t mlen,
                           uint64_t sample_ts, uint64_t sys_ts, const msd_hit *hits, uint64_t nhits,
                           uint64_t *hi, const msd_try *tries, msd_emit_fn emit, void *user)
{
    msd_stats *st = r->stats;
    (void)batch_chunk0;
    const uint64_t base = (uint64_t)b * MSD_CHUNK_SAMPLES; /* batch-relative */
    const uint64_t end = base + mlen;
    uint64_t resume = base; /* first position not covered by a skip-ahead */

    r->ifile_now = sys_ts; /* demod_2400.c:252-255 */

    for (; *hi < nhits; ++*hi) {
        const msd_hit h = hits[*hi];
        const uint64_t a = MSD_HIT_POS(h);
        if (a >= end)
            break;
        if (a < resume)
            continue; /* inside the previous message (demod_2400.c:416) */

        const unsigned mask = MSD_HIT_MASK(h);
        if (mask & 1) { st->demod_preamblePhase[0]++; st->demod_preamblePhase[1]++; }
        if (mask & 2) { st->demod_preamblePhase[2]++; st->demod_preamblePhase[3]++; }
        if (mask & 4) { st->demod_preamblePhase[4]++; }
        st->demod_preambles++;

        /* best phase: strict '>' so the first-tried phase wins ties (demod_2400.c:218); every
         * try that is not in the list scores -2 whatever the filter holds */
        const unsigned nlive = MSD_HIT_NLIVE(h);
        const msd_try *t = tries + MSD_HIT_TRY(h);
        int bestscore = -2, known_best = 0;
        const msd_try *best = 0;
        for (unsigned k = 0; k < nlive; ++k) {
            int known = filter_test(&r->filter, t[k].addr);
            int s = score_try(&t[k], known);
            if (s > bestscore) {
                bestscore = s;
                best = &t[k];
                known_best = known;
            }
        }
        if (bestscore < 0) {
            if (bestscore == -1)
                st->demod_rejected_unknown_icao++;
            else
                st->demod_rejected_bad++;
            continue;
        }

        const uint32_t j = (uint32_t)(a - base);
        const int df = best->msg[0] >> 3;
        const int msgbits = (df & 0x10) ? 112 : 56;

        msd_message mm;
        memset(&mm, 0, sizeof mm);
        mm.timestampMsg = sample_ts + (uint64_t)j * 5 + (8 + 56) * 12 + best->tp;
        mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
        r->ifile_now = mm.sysTimestampMsg; /* demod_2400.c:363-366, before decode */
        mm.score = bestscore;
        mm.bestphase = best->tp;

        /* acceptance part of decodeModesMessage (mode_s.c:424-555); the filter has not changed
         * since the score, so `known_best` is what its icaoFilterTest calls return */
        memcpy(mm.msg, best->msg, 14);
        mm.msgtype = (uint8_t)df;
        mm.msgbits = (uint8_t)msgbits;
        mm.crc = best->crc;
        const int nerr = (best->errbit != 0xff);
        int verdict = 0;
        switch (df) {
        case 11:
            mm.iid = (uint8_t)(mm.crc & 0x7f);
            if (nerr && !known_best)
                verdict = -1; /* mode_s.c:492-498 */
            break;
        case 17: case 18:
            if (nerr && best->errbit >= 8 && best->errbit <= 31 && !known_best)
                verdict = -1; /* mode_s.c:522-526: the fix changed AA */
            break;
        default:
            if (!known_best)
                verdict = -1; /* unreachable: such a try scores < 0 */
            break;
        }
        if (verdict < 0) {
            st->demod_rejected_unknown_icao++;
            continue;
        }
        if (nerr) {
            mm.correctedbits = 1;
            mm.msg[best->errbit >> 3] ^= (uint8_t)(0x80u >> (best->errbit & 7)); /* crc.c:417-425 */
        }
        mm.addr = best->addr; /* CRC for AP formats; AA after the fix otherwise (mode_s.c:559-562) */
        if (!nerr && (df == 17 || (df == 11 && mm.iid == 0)))
            filter_add(&r->filter, mm.addr); /* mode_s.c:717-726 */

        st->demod_accepted[mm.correctedbits]++;
        st->demod_bestPhase[best->tp - 4]++;

        const int signal_len = msgbits * 12 / 5;
        resume = a + (uint64_t)signal_len + 1; /* j += len (demod_2400.c:416), then the loop's ++ */
        emit(&mm, (a << 16) | (uint64_t)signal_len, b, user);
    }
}

/* The skip-ahead part of demodulate2400AC (demod_2400.c:522-708): every candidate in `ac` has
 * already passed all level/bit tests on the GPU. */
static void resolve_mode_ac(msd_resolver *r, uint32_t b, uint32_t mlen, uint64_t sample_ts,
                            uint64_t sys_ts, const msd_ac_hit *ac, uint64_t nac, uint64_t *ai,
                            msd_emit_fn emit, void *user)
{
    const uint64_t base = (uint64_t)b * MSD_CHUNK_SAMPLES; /* batch-relative, like msd_ac_hit.pos */
    const uint64_t end = base + mlen;
    uint64_t resume = base;
    for (; *ai < nac; ++*ai) {
        const msd_ac_hit *c = &ac[*ai];
        if (c->pos >= end)
            break;
        if (c->pos < resume)
            continue;
        msd_message mm;
        memset(&mm, 0, sizeof mm);
        mm.timestampMsg = sample_ts + c->f2_clock / 5; /* demod_2400.c:695 */
        mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
        mm.msgtype = 32; /* mode_ac.c:168-202 */
        mm.msgbits = 16;
        mm.msg[0] = (uint8_t)(c->modeac >> 8);
        mm.msg[1] = (uint8_t)c->modeac;
        mm.addr = (c->modeac & 0x0000FF7Fu) | (1u << 24);
        emit(&mm, 0, b, user);
        resume = c->pos + (20 * 87 / 25) + 1; /* demod_2400.c:705 plus the loop's ++ */
        r->stats->demod_modeac++;
    }
}

void msd_resolve_batch(msd_resolver *r, uint64_t first_chunk, uint32_t nbuffers,
                       const uint32_t *valid, const msd_hit *hits, uint64_t nhits,
                       const msd_try *tries, uint64_t ntries, const msd_ac_hit *ac, uint64_t nac,
                       const uint64_t *ts_override, msd_emit_fn emit, void *user)
{
    uint64_t hi = 0, ai = 0;
    (void)ntries;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        /* sdr_ifile.c:187-190 with startup_time = 0 */
        uint64_t sample_ts = (uint64_t)(r->sample_counter * 12e6 / 2400000.0);
        uint64_t sys_ts = sample_ts / 12000u;
        if (ts_override) {
            sample_ts = ts_override[2 * b];
            sys_ts = ts_override[2 * b + 1];
        }
        const uint32_t mlen = valid[b];

        resolve_mode_s(r, first_chunk, b, mlen, sample_ts, sys_ts, hits, nhits, &hi, tries, emit, user);
        if (r->mode_ac)
            resolve_mode_ac(r, b, mlen, sample_ts, sys_ts, ac, nac, &ai, emit, user);

        r->stats->samples_processed += (uint64_t)mlen + MSD_OVERLAP; /* readsb.c:835 */
        r->stats->buffers++;
        r->sample_counter += mlen;
        filter_expire(&r->filter, r->ifile_now); /* readsb.c:331, after the buffer */
    }
}

void msd_resolve_power(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const double *means,
                       msd_message *msgs, const uint64_t *power_req, const uint32_t *buffer,
                       const uint64_t *power, uint64_t nmsgs)
{
    msd_stats *st = r->stats;
    uint64_t i = 0;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        uint64_t sum_scaled_signal_power = 0;
        for (; i < nmsgs && buffer[i] == b; ++i) {
            if (!power_req[i])
                continue; /* Mode A/C */
            const int signal_len = (int)(power_req[i] & 0xffffu);
            const uint64_t scaled = power[i];
            /* demod_2400.c:386-408 */
            const double signal_power = scaled / 65535.0 / 65535.0;
            msgs[i].signalLevel = signal_power / signal_len;
            st->signal_power_sum += signal_power;
            st->signal_power_count += (uint64_t)signal_len;
            sum_scaled_signal_power += scaled;
            if (msgs[i].signalLevel > st->peak_signal_power)
                st->peak_signal_power = msgs[i].signalLevel;
            if (msgs[i].signalLevel > 0.50119)
                st->strong_signal_count++;
        }
        { /* demod_2400.c:422-427 */
            const uint32_t mlen = valid[b];
            const double sum_signal_power = sum_scaled_signal_power / 65535.0 / 65535.0;
            st->noise_power_sum += (means[2 * b + 1] * mlen - sum_signal_power);
            st->noise_power_count += mlen;
        }
    }
}
