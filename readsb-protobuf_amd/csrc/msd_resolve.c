/*
 * msd_resolve.c -- the ordered resolve stage: replays demodulate2400's sequential state machine
 * (skip-ahead, ICAO filter, ifile clock, counters) over the candidate lists the GPU produced.
 *
 * What is sequential here, and how it is still run on many host cores:
 *   - the skip-ahead after an accepted message never crosses a buffer boundary
 *     (demod_2400.c:257,416), so buffers are independent except for the ICAO filter;
 *   - the filter is only written by accepted clean DF17 / DF11(II=0) messages
 *     (mode_s.c:717-726) and aged between buffers (readsb.c:331), and its *membership* changes
 *     rarely (a new aircraft, or a 60 s flip that drops a silent one).
 * So a batch is resolved speculatively: every buffer is resolved in parallel against a snapshot
 * of the filter (plus the addresses the buffer itself adds); a cheap sequential pass then replays
 * the adds and flips, giving every buffer the membership "version" it should have seen; buffers
 * resolved against another version are redone.  The earliest stale buffer always has correct
 * inputs, so this converges to exactly the sequential result (usually in one or two passes).
 *
 * Host C on purpose (the north star keeps the host in C): the data-parallel work -- IQ->magnitude,
 * preamble tests, bit slicing, CRC, syndrome lookup, signal power -- is done on the GPU.
 */
#define _POSIX_C_SOURCE 200809L
#include "msd_internal.h"
#include "modes_hip.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>

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

static int table_has(const uint32_t *t, uint32_t addr, uint32_t start);

static void filter_init(msd_filter *f) /* icao_filter.c:67-71 */
{
    memset(f->slot, 0xFF, sizeof f->slot);
    f->active = 0;
    f->next_flip = 0;
    f->set_hash = 0;
    f->set_count = 0;
    f->active_used = 0;
}

/* exact comparison of two filters' membership (used to confirm a set-hash match) */
static int same_members(const msd_filter *x, const msd_filter *y)
{
    if (x->set_count != y->set_count)
        return 0;
    for (int w = 0; w < 2; ++w)
        for (uint32_t i = 0; i < SLOTS; ++i) {
            const uint32_t v = x->slot[w][i];
            if (v != VACANT && !(table_has(y->slot[0], v, hash24(v)) || table_has(y->slot[1], v, hash24(v))))
                return 0;
        }
    return 1; /* x is a subset of y and the sizes agree */
}

static int table_has(const uint32_t *t, uint32_t addr, uint32_t start)
{
    uint32_t h = start;
    while (t[h] != VACANT && t[h] != addr) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            break;
    }
    return t[h] == addr;
}

static int filter_test(const msd_filter *f, uint32_t addr) /* icao_filter.c:99-119 */
{
    const uint32_t start = hash24(addr);
    return table_has(f->slot[0], addr, start) || table_has(f->slot[1], addr, start);
}

static uint64_t mix_addr(uint32_t a)
{
    uint64_t z = (uint64_t)a + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/* slot index at which a probe for addr ends (its primary copy if present) */
static uint32_t probe_index(const uint32_t *t, uint32_t addr)
{
    const uint32_t start = hash24(addr);
    uint32_t h = start;
    while (t[h] != VACANT && t[h] != addr) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            break;
    }
    return h;
}

static int filter_add_raw(msd_filter *f, uint32_t addr);

/* icao_filter.c:76-97, plus bookkeeping of the membership (set hash and size).
 * Returns 1 if the set of known addresses grew. */
static int filter_add(msd_filter *f, uint32_t addr)
{
    /* already in the active table: both inserts are no-ops (entries are never removed one by one) */
    if (table_has(f->slot[f->active], addr, hash24(addr)))
        return 0;
    const int was_known = filter_test(f, addr);
    filter_add_raw(f, addr);
    if (!was_known && filter_test(f, addr)) {
        f->set_hash ^= mix_addr(addr);
        f->set_count++;
        return 1;
    }
    return 0;
}

static int filter_add_raw(msd_filter *f, uint32_t addr)
{
    uint32_t *t = f->slot[f->active];
    uint32_t start = hash24(addr), h = start;
    while (t[h] != VACANT && t[h] != addr) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            return 0; /* table full: the reference gives up on both inserts */
    }
    if (t[h] == VACANT) {
        t[h] = addr;
        f->active_used++;
    }

    /* second copy keyed by the low 16 bits (icao_filter.c:89-97) */
    const uint32_t low = addr & 0xffffu;
    start = h = hash24(low);
    while (t[h] != VACANT && (t[h] & 0xffffu) != low) {
        h = (h + 1) & (SLOTS - 1);
        if (h == start)
            return 0;
    }
    if (t[h] == VACANT) {
        t[h] = addr;
        f->active_used++;
    }
    return 1;
}

/* icao_filter.c:150-164.  Returns 1 if the flip dropped an address (membership shrank). */
static int filter_expire(msd_filter *f, uint64_t now)
{
    if (now < f->next_flip)
        return 0;
    const int other = f->active ^ 1;
    int dropped = 0;
    const uint32_t *gone = f->slot[other], *stay = f->slot[f->active];
    for (uint32_t i = 0; i < SLOTS; ++i) {
        const uint32_t v = gone[i];
        /* every address is stored twice (icao_filter.c:89-97): count its primary copy only */
        if (v != VACANT && probe_index(gone, v) == i && !table_has(stay, v, hash24(v))) {
            f->set_hash ^= mix_addr(v);
            f->set_count--;
            dropped = 1;
        }
    }
    memset(f->slot[other], 0xFF, sizeof f->slot[other]);
    f->active = other;
    f->active_used = 0;
    f->next_flip = now + 60000u;
    return dropped;
}

/* ---------------------------------------------------------------------------------------- */
/* the addresses one buffer has added so far (consulted on top of the snapshot)             */
/* ---------------------------------------------------------------------------------------- */

#define MAX_SPECULATIVE_PASSES 4u
#define LOCAL_SLOTS 4096u /* a buffer holds at most 131072/135 < 1024 accepted messages */

typedef struct local_set {
    uint32_t slot[LOCAL_SLOTS];
    uint32_t n;
} local_set;

static void local_init(local_set *s)
{
    memset(s->slot, 0xFF, sizeof s->slot);
    s->n = 0;
}

static int local_has(const local_set *s, uint32_t addr)
{
    if (!s->n)
        return 0;
    uint32_t h = (addr * 2654435761u) >> 20;
    while (s->slot[h] != VACANT) {
        if (s->slot[h] == addr)
            return 1;
        h = (h + 1) & (LOCAL_SLOTS - 1);
    }
    return 0;
}

static int local_add(local_set *s, uint32_t addr) /* returns 1 if newly added */
{
    uint32_t h = (addr * 2654435761u) >> 20;
    while (s->slot[h] != VACANT) {
        if (s->slot[h] == addr)
            return 0;
        h = (h + 1) & (LOCAL_SLOTS - 1);
    }
    s->slot[h] = addr;
    s->n++;
    return 1;
}

/* ---------------------------------------------------------------------------------------- */
/* thread pool                                                                              */
/* ---------------------------------------------------------------------------------------- */

typedef void (*pool_fn)(void *arg, uint32_t index);

struct msd_pool {
    pthread_t *threads;
    int nthreads; /* workers besides the caller */
    pthread_mutex_t mu;
    pthread_cond_t start, done;
    uint64_t generation;
    int running, shutdown;
    pool_fn fn;
    void *arg;
    uint32_t count;
    atomic_uint next;
};

static void pool_drain(struct msd_pool *p)
{
    for (;;) {
        const uint32_t i = atomic_fetch_add(&p->next, 1u);
        if (i >= p->count)
            break;
        p->fn(p->arg, i);
    }
}

static void *pool_main(void *arg)
{
    struct msd_pool *p = arg;
    uint64_t seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->shutdown && p->generation == seen)
            pthread_cond_wait(&p->start, &p->mu);
        if (p->shutdown)
            break;
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        pool_drain(p);
        pthread_mutex_lock(&p->mu);
        if (--p->running == 0)
            pthread_cond_signal(&p->done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static struct msd_pool *pool_create(int nworkers)
{
    struct msd_pool *p = calloc(1, sizeof *p);
    if (!p)
        return NULL;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->start, NULL);
    pthread_cond_init(&p->done, NULL);
    p->threads = calloc((size_t)(nworkers > 0 ? nworkers : 1), sizeof p->threads[0]);
    for (int i = 0; i < nworkers; ++i) {
        if (pthread_create(&p->threads[p->nthreads], NULL, pool_main, p) != 0)
            break;
        p->nthreads++;
    }
    return p;
}

static void pool_run(struct msd_pool *p, pool_fn fn, void *arg, uint32_t count)
{
    if (!p || p->nthreads == 0 || count < 2) {
        for (uint32_t i = 0; i < count; ++i)
            fn(arg, i);
        return;
    }
    pthread_mutex_lock(&p->mu);
    p->fn = fn;
    p->arg = arg;
    p->count = count;
    atomic_store(&p->next, 0u);
    p->running = p->nthreads;
    p->generation++;
    pthread_cond_broadcast(&p->start);
    pthread_mutex_unlock(&p->mu);
    pool_drain(p); /* the caller works too */
    pthread_mutex_lock(&p->mu);
    while (p->running)
        pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

static void pool_destroy(struct msd_pool *p)
{
    if (!p)
        return;
    pthread_mutex_lock(&p->mu);
    p->shutdown = 1;
    pthread_cond_broadcast(&p->start);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->nthreads; ++i)
        pthread_join(p->threads[i], NULL);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->start);
    pthread_cond_destroy(&p->done);
    free(p->threads);
    free(p);
}

/* ---------------------------------------------------------------------------------------- */
/* one buffer                                                                               */
/* ---------------------------------------------------------------------------------------- */

enum { C_PREAMBLES, C_BAD, C_UNKNOWN, C_ACC0, C_ACC1, C_ACC2, C_PPHASE0, C_BPHASE0 = C_PPHASE0 + 5,
       C_MODEAC = C_BPHASE0 + 5, C_COUNT };

typedef struct buf_result {
    uint32_t version_used;
    uint64_t end_now; /* Modes.ifile_now when the buffer is done */
    uint64_t ctr[C_COUNT];
    msd_message *msgs;
    uint64_t *reqs;
    uint32_t nmsgs, cap_msgs;
    uint32_t *adds; /* unique addresses passed to icaoFilterAdd, in order */
    uint32_t nadds, cap_adds;
} buf_result;

struct msd_batch_state {
    struct msd_pool *pool;
    int pool_tried;
    buf_result *res;
    uint32_t res_cap;
    msd_filter *snaps; /* membership versions of the filter within the current batch */
    uint32_t nsnaps, cap_snaps;
    uint32_t *want;    /* version each buffer should see */
    uint64_t *hit_begin, *ac_begin, *ts; /* per buffer */
    /* inputs of the running batch */
    const uint32_t *valid;
    const msd_hit *hits;
    uint64_t nhits;
    const msd_try *tries;
    const msd_ac_hit *ac;
    uint64_t nac;
    int mode_ac;
    uint32_t *todo;
    uint32_t ntodo;
    msd_filter work; /* GPU resolve: the filter behind the last replayed buffer */
    uint32_t *short_ok, cap_short;
    uint32_t *pmap, pmap_cap, pmap_alloc; /* address -> 1 + index in the prediction list */
    uint8_t *pconf, *stale;
    uint32_t pconf_cap, stale_cap;
    uint32_t undo_idx[2 * MSD_PRED_LIST], undo_first[2 * MSD_PRED_LIST]; /* to take back corrections on -2 */
};

static void push_msg(buf_result *br, const msd_message *mm, uint64_t req)
{
    if (br->nmsgs == br->cap_msgs) {
        const uint32_t cap = br->cap_msgs ? br->cap_msgs * 2 : 64;
        br->msgs = realloc(br->msgs, (size_t)cap * sizeof br->msgs[0]);
        br->reqs = realloc(br->reqs, (size_t)cap * sizeof br->reqs[0]);
        br->cap_msgs = cap;
    }
    br->msgs[br->nmsgs] = *mm;
    br->reqs[br->nmsgs] = req;
    br->nmsgs++;
}

static void push_add(buf_result *br, uint32_t addr)
{
    if (br->nadds == br->cap_adds) {
        const uint32_t cap = br->cap_adds ? br->cap_adds * 2 : 64;
        br->adds = realloc(br->adds, (size_t)cap * sizeof br->adds[0]);
        br->cap_adds = cap;
    }
    br->adds[br->nadds++] = addr;
}

/* scoreModesMessage (mode_s.c:311-409) given what the GPU already derived for this try */
static int score_try(const msd_try *t, int known)
{
    const int df = t->msg[0] >> 3;
    const int nerr = (t->errbit != 0xff) + (t->errbit2 != 0xff);
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

/* demodulate2400 (demod_2400.c:236-428) for buffer b, everything except the signal-power
 * bookkeeping (msd_resolve_power), then the skip-ahead part of demodulate2400AC (:522-708). */
/* While the active table has room, "the filter as it was when the buffer started, plus the
 * addresses this buffer added" is exactly what the reference's filter answers inside the buffer.
 * icaoFilterAdd silently gives up on a full table (icao_filter.c:82-86), though, so when the table
 * could fill up within a buffer -- thousands of aircraft in one minute -- the buffer is replayed
 * with `live`: adds go to the filter at once and only the filter is asked. */
#define FULL_GUARD 6000u /* occupied slots; a buffer adds at most 970 addresses, two slots each */

static void resolve_buffer(const struct msd_batch_state *bs, uint32_t b, const msd_filter *snap,
                           uint32_t version, buf_result *br, msd_filter *live)
{
    local_set local;
    local_init(&local);
    memset(br->ctr, 0, sizeof br->ctr);
    br->nmsgs = 0;
    br->nadds = 0;
    br->version_used = version;

    const uint32_t mlen = bs->valid[b];
    const uint64_t sample_ts = bs->ts[2 * b], sys_ts = bs->ts[2 * b + 1];
    const uint64_t base = (uint64_t)b * MSD_CHUNK_SAMPLES; /* batch-relative */
    const uint64_t end = base + mlen;
    uint64_t resume = base; /* first position not covered by a skip-ahead */
    uint64_t now = sys_ts;  /* demod_2400.c:252-255 */

    const msd_hit *hits = bs->hits;
    const msd_try *tries = bs->tries;
    for (uint64_t hi = bs->hit_begin[b]; hi < bs->nhits; ++hi) {
        const msd_hit h = hits[hi];
        const uint64_t a = MSD_HIT_POS(h);
        if (a >= end)
            break;
        if (a < resume)
            continue; /* inside the previous message (demod_2400.c:416) */

        const unsigned mask = MSD_HIT_MASK(h);
        if (mask & 1) { br->ctr[C_PPHASE0 + 0]++; br->ctr[C_PPHASE0 + 1]++; }
        if (mask & 2) { br->ctr[C_PPHASE0 + 2]++; br->ctr[C_PPHASE0 + 3]++; }
        if (mask & 4) { br->ctr[C_PPHASE0 + 4]++; }
        br->ctr[C_PREAMBLES]++;

        /* best phase: strict '>' so the first-tried phase wins ties (demod_2400.c:218); every
         * try that is not in the list scores -2 whatever the filter holds */
        const unsigned nlive = MSD_HIT_NLIVE(h);
        const msd_try *t = tries + MSD_HIT_TRY(h);
        int bestscore = -2, known_best = 0;
        const msd_try *best = 0;
        for (unsigned k = 0; k < nlive; ++k) {
            const int known = live ? filter_test(live, t[k].addr)
                                   : filter_test(snap, t[k].addr) || local_has(&local, t[k].addr);
            const int s = score_try(&t[k], known);
            if (s > bestscore) {
                bestscore = s;
                best = &t[k];
                known_best = known;
            }
        }
        if (bestscore < 0) {
            br->ctr[bestscore == -1 ? C_UNKNOWN : C_BAD]++;
            continue;
        }

        const uint32_t j = (uint32_t)(a - base);
        const int df = best->msg[0] >> 3;
        const int msgbits = (df & 0x10) ? 112 : 56;

        msd_message mm;
        memset(&mm, 0, sizeof mm);
        mm.timestampMsg = sample_ts + (uint64_t)j * 5 + (8 + 56) * 12 + best->tp;
        mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
        now = mm.sysTimestampMsg; /* demod_2400.c:363-366, before decode */
        mm.score = bestscore;
        mm.bestphase = best->tp;

        /* acceptance part of decodeModesMessage (mode_s.c:424-555); the filter has not changed
         * since the score, so `known_best` is what its icaoFilterTest calls return */
        memcpy(mm.msg, best->msg, 14);
        mm.msgtype = (uint8_t)df;
        mm.msgbits = (uint8_t)msgbits;
        mm.crc = best->crc;
        const int nerr = (best->errbit != 0xff) + (best->errbit2 != 0xff);
        const int fix_in_aa = (best->errbit >= 8 && best->errbit <= 31) || (best->errbit2 >= 8 && best->errbit2 <= 31);
        int verdict = 0;
        switch (df) {
        case 11:
            mm.iid = (uint8_t)(mm.crc & 0x7f);
            if (nerr && !known_best)
                verdict = -1; /* mode_s.c:492-498 */
            break;
        case 17: case 18:
            if (nerr && fix_in_aa && !known_best)
                verdict = -1; /* mode_s.c:522-526: the fix changed AA */
            break;
        default:
            if (!known_best)
                verdict = -1; /* unreachable: such a try scores < 0 */
            break;
        }
        if (verdict < 0) {
            br->ctr[C_UNKNOWN]++;
            continue;
        }
        if (nerr) {
            mm.correctedbits = (uint8_t)nerr;
            mm.msg[best->errbit >> 3] ^= (uint8_t)(0x80u >> (best->errbit & 7)); /* crc.c:417-425 */
            if (nerr > 1)
                mm.msg[best->errbit2 >> 3] ^= (uint8_t)(0x80u >> (best->errbit2 & 7));
        }
        mm.addr = best->addr; /* CRC for AP formats; AA after the fix otherwise (mode_s.c:559-562) */
        if (!nerr && (df == 17 || (df == 11 && mm.iid == 0))) { /* mode_s.c:717-726 */
            if (live)
                filter_add(live, mm.addr); /* the caller must not apply br->adds again (there are none) */
            else if (local_add(&local, mm.addr))
                push_add(br, mm.addr);
        }

        br->ctr[C_ACC0 + mm.correctedbits]++;
        br->ctr[C_BPHASE0 + best->tp - 4]++;

        const int signal_len = msgbits * 12 / 5;
        resume = a + (uint64_t)signal_len + 1; /* j += len (demod_2400.c:416), then the loop's ++ */
        push_msg(br, &mm, (a << 16) | (uint64_t)signal_len);
    }
    br->end_now = now;

    if (bs->mode_ac) {
        resume = base;
        for (uint64_t ai = bs->ac_begin[b]; ai < bs->nac; ++ai) {
            const msd_ac_hit *c = &bs->ac[ai];
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
            push_msg(br, &mm, 0);
            resume = c->pos + (20 * 87 / 25) + 1; /* demod_2400.c:705 plus the loop's ++ */
            br->ctr[C_MODEAC]++;
        }
    }
}

static void job_resolve(void *arg, uint32_t index)
{
    struct msd_batch_state *bs = arg;
    const uint32_t b = bs->todo[index];
    const uint32_t v = bs->want[b];
    resolve_buffer(bs, b, &bs->snaps[v], v, &bs->res[b], NULL);
}

/* ---------------------------------------------------------------------------------------- */

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int default_threads(void)
{
    /* an eighth of the host's CPUs (an 8-GPU node runs 8 contexts), between 4 and 64 */
    long n = sysconf(_SC_NPROCESSORS_ONLN) / 8;
    return (int)(n < 4 ? 4 : (n > 64 ? 64 : n));
}

/* a new capture: empty filter, clocks at zero ... */
void msd_resolver_reset_state(msd_resolver *r)
{
    filter_init(&r->filter);
    r->ifile_now = 0;
    r->sample_counter = 0;
}

/* ... and its counters */
void msd_resolver_reset_stats(msd_resolver *r)
{
    if (r->stats)
        memset(r->stats, 0, sizeof *r->stats);
}

void msd_resolver_reset(msd_resolver *r)
{
    msd_resolver_reset_state(r);
    msd_resolver_reset_stats(r);
}

void msd_resolver_free(msd_resolver *r)
{
    struct msd_batch_state *bs = r->batch;
    if (!bs)
        return;
    pool_destroy(bs->pool);
    for (uint32_t i = 0; i < bs->res_cap; ++i) {
        free(bs->res[i].msgs);
        free(bs->res[i].reqs);
        free(bs->res[i].adds);
    }
    free(bs->res);
    free(bs->snaps);
    free(bs->want);
    free(bs->hit_begin);
    free(bs->ac_begin);
    free(bs->ts);
    free(bs->todo);
    free(bs->short_ok);
    free(bs->pmap);
    free(bs->pconf);
    free(bs->stale);
    free(bs);
    r->batch = NULL;
}

static struct msd_batch_state *batch_state(msd_resolver *r, uint32_t nbuffers)
{
    struct msd_batch_state *bs = r->batch;
    if (!bs) {
        bs = calloc(1, sizeof *bs);
        if (!bs)
            return NULL;
        r->batch = bs;
    }
    if (nbuffers > bs->res_cap) {
        bs->res = realloc(bs->res, (size_t)nbuffers * sizeof bs->res[0]);
        memset(bs->res + bs->res_cap, 0, (size_t)(nbuffers - bs->res_cap) * sizeof bs->res[0]);
        bs->want = realloc(bs->want, (size_t)nbuffers * sizeof bs->want[0]);
        bs->hit_begin = realloc(bs->hit_begin, (size_t)nbuffers * sizeof bs->hit_begin[0]);
        bs->ac_begin = realloc(bs->ac_begin, (size_t)nbuffers * sizeof bs->ac_begin[0]);
        bs->ts = realloc(bs->ts, (size_t)nbuffers * 2 * sizeof bs->ts[0]);
        bs->todo = realloc(bs->todo, (size_t)nbuffers * sizeof bs->todo[0]);
        bs->res_cap = nbuffers;
    }
    return bs;
}

/* index of a stored snapshot with exactly this membership, adding one if there is none; ids stay
 * valid for the whole batch, so "resolved against version v" can be compared across passes */
static uint32_t push_snapshot(struct msd_batch_state *bs, const msd_filter *f)
{
    for (uint32_t i = 0; i < bs->nsnaps; ++i)
        if (bs->snaps[i].set_hash == f->set_hash && same_members(&bs->snaps[i], f))
            return i;
    if (bs->nsnaps == bs->cap_snaps) {
        const uint32_t cap = bs->cap_snaps ? bs->cap_snaps * 2 : 4;
        bs->snaps = realloc(bs->snaps, (size_t)cap * sizeof bs->snaps[0]);
        bs->cap_snaps = cap;
    }
    bs->snaps[bs->nsnaps] = *f;
    return bs->nsnaps++;
}

static uint64_t lower_bound_hit(const msd_hit *hits, uint64_t n, uint64_t pos)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (MSD_HIT_POS(hits[mid]) < pos)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

static uint64_t lower_bound_ac(const msd_ac_hit *ac, uint64_t n, uint64_t pos)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ac[mid].pos < pos)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

void msd_resolve_batch(msd_resolver *r, uint64_t first_chunk, uint32_t nbuffers,
                       const uint32_t *valid, const msd_hit *hits, uint64_t nhits,
                       const msd_try *tries, uint64_t ntries, const msd_ac_hit *ac, uint64_t nac,
                       const uint64_t *ts_override, msd_emit_fn emit, void *user)
{
    (void)first_chunk;
    (void)ntries;
    if (nbuffers == 0)
        return;
    struct msd_batch_state *bs = batch_state(r, nbuffers);
    if (!bs)
        return;
    bs->valid = valid;
    bs->hits = hits;
    bs->nhits = nhits;
    bs->tries = tries;
    bs->ac = ac;
    bs->nac = nac;
    bs->mode_ac = r->mode_ac;

    { /* per-buffer clocks (sdr_ifile.c:187-190, startup_time = 0) and candidate ranges */
        uint64_t counter = r->sample_counter;
        for (uint32_t b = 0; b < nbuffers; ++b) {
            uint64_t sample_ts = (uint64_t)(counter * 12e6 / 2400000.0);
            uint64_t sys_ts = sample_ts / 12000u;
            if (ts_override) {
                sample_ts = ts_override[2 * b];
                sys_ts = ts_override[2 * b + 1];
            }
            bs->ts[2 * b] = sample_ts;
            bs->ts[2 * b + 1] = sys_ts;
            counter += valid[b];
            bs->hit_begin[b] = lower_bound_hit(hits, nhits, (uint64_t)b * MSD_CHUNK_SAMPLES);
            bs->ac_begin[b] = ac ? lower_bound_ac(ac, nac, (uint64_t)b * MSD_CHUNK_SAMPLES) : 0;
        }
    }

    if (!bs->pool && !bs->pool_tried && nbuffers >= 4) { /* the worker threads exist only once the host path is used */
        const int n = r->threads > 0 ? r->threads : default_threads();
        bs->pool = pool_create(n - 1);
        bs->pool_tried = 1;
    }
    const int serial = !bs->pool || bs->pool->nthreads == 0 || nbuffers < 4;
    msd_filter work;
    const int trace = r->trace;
    double t_par = 0, t_seq = 0, t0 = trace ? now_ms() : 0;
    uint32_t npass = 0;
    if (serial) {
        /* plain sequential replay: the live filter is the snapshot and a buffer's adds are applied
         * when it is done -- equal to the reference because, inside a buffer, the filter is only
         * read through "snapshot or this buffer's own adds" */
        work = r->filter;
        for (uint32_t b = 0; b < nbuffers; ++b) {
            buf_result *br = &bs->res[b];
            resolve_buffer(bs, b, &work, 0, br, work.active_used > FULL_GUARD ? &work : NULL);
            for (uint32_t i = 0; i < br->nadds; ++i)
                filter_add(&work, br->adds[i]);
            filter_expire(&work, br->end_now); /* readsb.c:331, after the buffer */
        }
    } else {
        bs->nsnaps = 0;
        push_snapshot(bs, &r->filter);
        for (uint32_t b = 0; b < nbuffers; ++b) {
            bs->want[b] = 0;
            bs->todo[b] = b;
        }
        bs->ntodo = nbuffers;
        for (uint32_t pass = 0;; ++pass) {
            int force_tail = 0;
            double ta = trace ? now_ms() : 0;
            pool_run(bs->pool, job_resolve, bs, bs->ntodo);
            double tb = trace ? now_ms() : 0;
            t_par += tb - ta;
            ++npass;
            /* replay adds and flips in order; find the membership version every buffer must see */
            work = r->filter;
            uint32_t version = 0, first_stale = nbuffers;
            bs->ntodo = 0;
            for (uint32_t b = 0; b < nbuffers; ++b) {
                buf_result *br = &bs->res[b];
                bs->want[b] = version;
                if (br->version_used != version) {
                    if (first_stale == nbuffers)
                        first_stale = b;
                    bs->todo[bs->ntodo++] = b;
                }
                if (first_stale == nbuffers && work.active_used > FULL_GUARD) {
                    /* the active table may fill up: from here on only the exact sequential replay will do */
                    first_stale = b;
                    if (bs->ntodo == 0 || bs->todo[bs->ntodo - 1] != b)
                        bs->todo[bs->ntodo++] = b;
                    force_tail = 1;
                    break;
                }
                if (pass >= MAX_SPECULATIVE_PASSES && first_stale != nbuffers)
                    break; /* `work` is now the exact state in front of the first stale buffer */
                int changed = 0;
                for (uint32_t i = 0; i < br->nadds; ++i)
                    changed |= filter_add(&work, br->adds[i]);
                changed |= filter_expire(&work, br->end_now);
                if (changed && b + 1 < nbuffers)
                    version = push_snapshot(bs, &work);
            }
            if (bs->ntodo == 0)
                break;
            if (pass >= MAX_SPECULATIVE_PASSES || force_tail) {
                /* membership keeps changing (every pass is exact up to its first stale buffer, but
                 * an input whose adds shift from pass to pass would need one pass per buffer):
                 * finish the tail with the plain sequential replay */
                for (uint32_t b = first_stale; b < nbuffers; ++b) {
                    buf_result *br = &bs->res[b];
                    resolve_buffer(bs, b, &work, 0, br, work.active_used > FULL_GUARD ? &work : NULL);
                    for (uint32_t i = 0; i < br->nadds; ++i)
                        filter_add(&work, br->adds[i]);
                    filter_expire(&work, br->end_now);
                }
                break;
            }
        }
    }

    if (trace) {
        t_seq = now_ms() - t0 - t_par;
        fprintf(stderr, "resolve: %u buffers %u passes parallel %.3f ms, setup+replay %.3f ms\n", nbuffers, npass, t_par, t_seq);
    }
    const double tc = trace ? now_ms() : 0;
    /* commit, in order */
    msd_stats *st = r->stats;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        const buf_result *br = &bs->res[b];
        st->demod_preambles += br->ctr[C_PREAMBLES];
        st->demod_rejected_bad += br->ctr[C_BAD];
        st->demod_rejected_unknown_icao += br->ctr[C_UNKNOWN];
        for (int k = 0; k < 3; ++k)
            st->demod_accepted[k] += br->ctr[C_ACC0 + k];
        for (int k = 0; k < 5; ++k) {
            st->demod_preamblePhase[k] += br->ctr[C_PPHASE0 + k];
            st->demod_bestPhase[k] += br->ctr[C_BPHASE0 + k];
        }
        st->demod_modeac += br->ctr[C_MODEAC];
        st->samples_processed += (uint64_t)valid[b] + MSD_OVERLAP; /* readsb.c:835 */
        st->buffers++;
        r->sample_counter += valid[b];
        if (br->nmsgs)
            emit(br->msgs, br->reqs, br->nmsgs, b, user);
    }
    r->filter = work;
    r->ifile_now = bs->res[nbuffers - 1].end_now;
    if (trace)
        fprintf(stderr, "resolve: commit %.3f ms\n", now_ms() - tc);
}

/* ---------------------------------------------------------------------------------------- */
/* host half of the GPU resolve: only the cross-buffer replay stays here                     */
/* ---------------------------------------------------------------------------------------- */

void msd_gpu_resolve_begin(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, uint64_t *ts,
                           uint32_t *snap_idx, uint32_t *todo, uint32_t *ntodo)
{
    struct msd_batch_state *bs = batch_state(r, nbuffers);
    uint64_t counter = r->sample_counter;
    for (uint32_t b = 0; b < nbuffers; ++b) { /* sdr_ifile.c:187-190, startup_time = 0 */
        const uint64_t sample_ts = (uint64_t)(counter * 12e6 / 2400000.0);
        ts[2 * b] = sample_ts;
        ts[2 * b + 1] = sample_ts / 12000u;
        counter += valid[b];
        snap_idx[b] = 0;
        todo[b] = b;
    }
    *ntodo = nbuffers;
    bs->nsnaps = 0;
    push_snapshot(bs, &r->filter);
}

uint32_t msd_gpu_resolve_nsnaps(const msd_resolver *r)
{
    return r->batch ? r->batch->nsnaps : 0;
}

const uint32_t *msd_gpu_resolve_snapshot(const msd_resolver *r, uint32_t index, uint32_t *active)
{
    *active = (uint32_t)r->batch->snaps[index].active;
    return &r->batch->snaps[index].slot[0][0];
}

/* every address of x's active table is in y's */
static int active_subset(const msd_filter *x, const msd_filter *y)
{
    const uint32_t *xa = x->slot[x->active], *ya = y->slot[y->active];
    for (uint32_t i = 0; i < SLOTS; ++i)
        if (xa[i] != VACANT && !table_has(ya, xa[i], hash24(xa[i])))
            return 0;
    return 1;
}

/* index of addr in the batch's prediction list, -1 if it is not there */
static int pred_find(const struct msd_batch_state *bs, const msd_pred_entry *pred, uint32_t addr)
{
    if (!bs->pmap_cap)
        return -1;
    uint32_t h = (uint32_t)mix_addr(addr) & (bs->pmap_cap - 1);
    while (bs->pmap[h]) {
        if (pred[bs->pmap[h] - 1].addr == addr)
            return (int)bs->pmap[h] - 1;
        h = (h + 1) & (bs->pmap_cap - 1);
    }
    return -1;
}

int msd_gpu_resolve_replay(msd_resolver *r, uint32_t nbuffers, const msd_rbuf *rb, const uint32_t *all_adds,
                           uint32_t inline_adds, uint32_t pass, uint32_t max_snaps, msd_pred_entry *pred, uint32_t npred,
                           msd_pred_patch *patches, uint32_t *npatches, uint32_t *snap_idx, uint32_t *todo,
                           uint32_t *ntodo)
{
    struct msd_batch_state *bs = r->batch;
    const int trace_replay = r->trace;
    *npatches = 0;
    if (npred > MSD_PRED_LIST)
        return -1; /* thousands of new aircraft in one batch: the table overflowed */
    bs->work = r->filter;
    uint32_t version = 0, n = 0, np = 0;
    /* short_ok[v]: 1 + the flip count at which "snapshot v's active table is a subset of the live
     * one" was last verified; it stays true until the next flip (active tables only grow) */
    uint32_t flips = 0;
    if (bs->cap_snaps > bs->cap_short) {
        bs->short_ok = realloc(bs->short_ok, (size_t)bs->cap_snaps * sizeof bs->short_ok[0]);
        bs->cap_short = bs->cap_snaps;
    }
    memset(bs->short_ok, 0, (size_t)bs->cap_short * sizeof bs->short_ok[0]);
    { /* addr -> prediction, confirmation flags, stale marks */
        uint32_t cap = 64;
        while (cap < 2 * npred + 2)
            cap *= 2;
        if (cap > bs->pmap_alloc) {
            bs->pmap = realloc(bs->pmap, (size_t)cap * sizeof bs->pmap[0]);
            bs->pmap_alloc = cap;
        }
        bs->pmap_cap = cap;
        memset(bs->pmap, 0, (size_t)cap * sizeof bs->pmap[0]);
        if (npred > bs->pconf_cap) {
            bs->pconf = realloc(bs->pconf, npred);
            bs->pconf_cap = npred;
        }
        if (npred)
            memset(bs->pconf, 0, npred);
        for (uint32_t i = 0; i < npred; ++i) {
            uint32_t h = (uint32_t)mix_addr(pred[i].addr) & (cap - 1);
            while (bs->pmap[h])
                h = (h + 1) & (cap - 1);
            bs->pmap[h] = i + 1;
            /* The scan kernel notes every address with a clean squitter, not only new ones (it does not know the
             * filter).  An address the filter holds as the batch begins needs no confirmation: its squitters add
             * nothing new, and "known from its first buffer on" is true as long as it stays a member -- if a flip
             * drops it first, the flip below takes the entry back like a confirmed one's. */
            if (pred[i].first != MSD_PRED_NEVER && filter_test(&bs->work, pred[i].addr))
                bs->pconf[i] = 2;
        }
        if (nbuffers > bs->stale_cap) {
            bs->stale = realloc(bs->stale, nbuffers);
            bs->stale_cap = nbuffers;
        }
        memset(bs->stale, 0, nbuffers);
    }
    for (uint32_t b = 0; b < nbuffers; ++b) {
        const msd_rbuf *br = &rb[b];
        if (br->fallback || bs->work.active_used > FULL_GUARD)
            return -1; /* a nearly full active table needs the exact sequential replay (resolve_buffer) */
        snap_idx[b] = version;
        if (br->version_used != version) {
            bs->stale[b] = 1;
            if (trace_replay)
                fprintf(stderr, "replay: pass %u buffer %u used snapshot %u, needs %u\n", pass, b, br->version_used, version);
        }
        const uint32_t v = br->version_used;
        int use_short = br->nshort <= inline_adds && v < bs->nsnaps;
        if (use_short && bs->short_ok[v] != flips + 1) {
            if (active_subset(&bs->snaps[v], &bs->work))
                bs->short_ok[v] = flips + 1;
            else
                use_short = 0;
        }
        if (!use_short && !all_adds && br->nadds) {
            while (np) { /* nothing may stick: the caller comes back with the complete lists */
                --np;
                pred[bs->undo_idx[np]].first = bs->undo_first[np];
            }
            return -2;
        }
        const uint32_t *adds = use_short ? br->adds : all_adds + (size_t)b * MSD_RB_MSG_CAP;
        const uint32_t nadds = use_short ? br->nshort : br->nadds;
        int regime_change = 0; /* the membership changed in a way the prediction table does not cover */
        for (uint32_t i = 0; i < nadds; ++i) {
            if (!filter_add(&bs->work, adds[i]))
                continue;
            /* a new member, known from buffer b + 1 on: was that predicted? */
            const int e = pred_find(bs, pred, adds[i]);
            if (trace_replay && !(e >= 0 && !bs->pconf[e] && pred[e].first == b))
                fprintf(stderr, "replay: pass %u buffer %u: new member %06x, prediction %s (first %d, confirmed %d)\n", pass, b,
                        adds[i], e < 0 ? "missing" : "off", e < 0 ? -1 : (int)pred[e].first, e < 0 ? 0 : bs->pconf[e]);
            if (e >= 0 && !bs->pconf[e] && pred[e].first <= b) {
                if (pred[e].first < b) { /* the predicted message was hidden: the buffers in between assumed too much */
                    for (uint32_t q = pred[e].first + 1; q <= b; ++q)
                        bs->stale[q] = 1;
                    bs->undo_idx[np] = (uint32_t)e;
                    bs->undo_first[np] = pred[e].first;
                    pred[e].first = b;
                    patches[np].slot = pred[e].slot;
                    patches[np++].first = b;
                }
                bs->pconf[e] = 1;
            } else {
                regime_change = 1;
            }
        }
        const int active_before = bs->work.active;
        if (filter_expire(&bs->work, br->end_now)) { /* readsb.c:331, after the buffer; members were dropped */
            regime_change = 1;
            for (uint32_t e = 0; e < npred; ++e) /* one of this batch's own additions? (needs a >60 s batch) */
                if (bs->pconf[e] && !filter_test(&bs->work, pred[e].addr)) {
                    bs->pconf[e] = 0;
                    bs->undo_idx[np] = e;
                    bs->undo_first[np] = pred[e].first;
                    pred[e].first = MSD_PRED_NEVER;
                    patches[np].slot = pred[e].slot;
                    patches[np++].first = MSD_PRED_NEVER;
                }
        }
        flips += bs->work.active != active_before;
        if (regime_change && b + 1 < nbuffers) {
            version = push_snapshot(bs, &bs->work);
            if (bs->nsnaps > max_snaps)
                return -1;
            if (bs->cap_snaps > bs->cap_short) { /* push_snapshot grew the array */
                bs->short_ok = realloc(bs->short_ok, (size_t)bs->cap_snaps * sizeof bs->short_ok[0]);
                memset(bs->short_ok + bs->cap_short, 0, (size_t)(bs->cap_snaps - bs->cap_short) * sizeof bs->short_ok[0]);
                bs->cap_short = bs->cap_snaps;
            }
        }
    }
    for (uint32_t e = 0; e < npred; ++e) /* predictions that never came true */
        if (!bs->pconf[e] && pred[e].first != MSD_PRED_NEVER) {
            if (trace_replay)
                fprintf(stderr, "replay: pass %u: %06x predicted for buffer %u was never added\n", pass, pred[e].addr, pred[e].first);
            for (uint32_t q = pred[e].first + 1; q < nbuffers; ++q)
                bs->stale[q] = 1;
            pred[e].first = MSD_PRED_NEVER;
            patches[np].slot = pred[e].slot;
            patches[np++].first = MSD_PRED_NEVER;
        }
    for (uint32_t b = 0; b < nbuffers; ++b)
        if (bs->stale[b])
            todo[n++] = b;
    *ntodo = n;
    *npatches = np;
    if (n == 0)
        return 0;
    return pass >= MAX_SPECULATIVE_PASSES ? -1 : 1;
}

/* What the next batch's resolve needs of a batch that went through: the filter as its last buffer left it, the
 * sample clock, Modes.ifile_now. */
void msd_gpu_resolve_commit_state(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const msd_rbuf *rb)
{
    for (uint32_t b = 0; b < nbuffers; ++b)
        r->sample_counter += valid[b];
    r->filter = r->batch->work;
    r->ifile_now = rb[nbuffers - 1].end_now;
}

/* ... and what only the statistics want (stats.h:61-80), added when the batch is delivered */
void msd_gpu_resolve_commit_stats(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const msd_rbuf *rb)
{
    msd_stats *st = r->stats;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        const msd_rbuf *br = &rb[b];
        st->demod_preambles += br->ctr[C_PREAMBLES];
        st->demod_rejected_bad += br->ctr[C_BAD];
        st->demod_rejected_unknown_icao += br->ctr[C_UNKNOWN];
        for (int k = 0; k < 3; ++k)
            st->demod_accepted[k] += br->ctr[C_ACC0 + k];
        for (int k = 0; k < 5; ++k) {
            st->demod_preamblePhase[k] += br->ctr[C_PPHASE0 + k];
            st->demod_bestPhase[k] += br->ctr[C_BPHASE0 + k];
        }
        st->demod_modeac += br->nac;
        st->samples_processed += (uint64_t)valid[b] + MSD_OVERLAP; /* readsb.c:835 */
        st->buffers++;
    }
}

void msd_resolve_power(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const double *means,
                       void *msgs_base, size_t msg_stride, const uint64_t *power_req, const uint32_t *buffer,
                       const void *power_base, size_t power_stride, uint64_t nmsgs)
{
    msd_stats *st = r->stats;
    uint64_t i = 0;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        uint64_t sum_scaled_signal_power = 0;
        for (; i < nmsgs && buffer[i] == b; ++i) {
            msd_message *mm = (msd_message *)((char *)msgs_base + i * msg_stride);
            if (power_req ? !power_req[i] : mm->msgtype == 32)
                continue; /* Mode A/C */
            const int signal_len = power_req ? (int)(power_req[i] & 0xffffu) : mm->msgbits * 12 / 5;
            uint64_t scaled; /* may sit in the bytes of mm->signalLevel (GPU resolve): no typed access */
            memcpy(&scaled, (const char *)power_base + i * power_stride, sizeof scaled);
            /* demod_2400.c:386-408 */
            const double signal_power = scaled / 65535.0 / 65535.0;
            mm->signalLevel = signal_power / signal_len;
            st->signal_power_sum += signal_power;
            st->signal_power_count += (uint64_t)signal_len;
            sum_scaled_signal_power += scaled;
            if (mm->signalLevel > st->peak_signal_power)
                st->peak_signal_power = mm->signalLevel;
            if (mm->signalLevel > 0.50119)
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

/* The statistics half alone, for the GPU resolve path (msd_capi.cpp): the record kernel has already written every
 * message's signalLevel (msd_emit_impl.h, the same double arithmetic), and the order-sensitive sums (one
 * dependent double add per message, demod_2400.c:398-408,422-427) follow from side[] = power sum | signal_len
 * << 48 (0: a Mode A/C reply) while the caller already works on the next batch. */
void msd_resolve_power_stats(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const double *means,
                             const uint32_t *buffer, const uint64_t *side, uint64_t nmsgs)
{
    msd_stats *st = r->stats;
    uint64_t i = 0;
    for (uint32_t b = 0; b < nbuffers; ++b) {
        uint64_t sum_scaled_signal_power = 0;
        for (; i < nmsgs && buffer[i] == b; ++i) {
            const int signal_len = (int)(side[i] >> 48);
            if (!signal_len)
                continue; /* Mode A/C */
            const uint64_t scaled = side[i] & 0xffffffffffffull;
            /* demod_2400.c:386-408 */
            const double signal_power = scaled / 65535.0 / 65535.0;
            const double level = signal_power / signal_len;
            st->signal_power_sum += signal_power;
            st->signal_power_count += (uint64_t)signal_len;
            sum_scaled_signal_power += scaled;
            if (level > st->peak_signal_power)
                st->peak_signal_power = level;
            if (level > 0.50119)
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
