/*
 * msd_fifo.c -- the magnitude-buffer queue between the reader thread and the demodulator, written
 * against the contract of fifo.h:34-120 (not against fifo.c): a producer acquires an unused buffer,
 * fills data[overlap..validLength) and enqueues it; the queue completes data[0..overlap) with the
 * last `overlap` samples of the buffer enqueued before it (zeros for the first one and after a
 * discontinuity); the consumer dequeues in order and releases.
 *
 * Layout: one allocation for all sample arrays, an array of buffer records, a ring of record
 * indices for the queued buffers (oldest first) and a stack of indices for the unused ones.  One
 * mutex, one condition variable that every state change broadcasts on; every waiter re-checks its
 * own predicate.  The overlap samples travel in a side array (`carry`), so no buffer ever refers to
 * another one.
 */
#define _POSIX_C_SOURCE 200809L
#include "modes_hip_readsb.h"

#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static struct {
    pthread_mutex_t mu;
    pthread_cond_t changed;
    struct msd_mag_buf *rec; /* [count] */
    uint16_t *samples;       /* [count][size] */
    unsigned count, size, overlap;
    unsigned *ring;          /* indices of the queued buffers */
    unsigned ring_first, ring_len;
    unsigned *unused;        /* stack of indices */
    unsigned unused_len;
    uint16_t *carry;         /* the newest enqueued buffer's last `overlap` samples */
    bool carry_valid;
    bool halted, ready;
} Q = {.mu = PTHREAD_MUTEX_INITIALIZER, .changed = PTHREAD_COND_INITIALIZER};

static void deadline_after(uint32_t ms, struct timespec *ts)
{
    clock_gettime(CLOCK_REALTIME, ts); /* pthread_cond_timedwait's default clock */
    ts->tv_sec += ms / 1000u;
    ts->tv_nsec += (long)(ms % 1000u) * 1000000L;
    if (ts->tv_nsec >= 1000000000L) {
        ts->tv_nsec -= 1000000000L;
        ts->tv_sec += 1;
    }
}

/* waits (mutex held) until *flag_a or *len is non-zero, or the deadline passes; ms == 0: no wait */
static void wait_for(const bool *stop, const unsigned *len, uint32_t ms)
{
    if (*stop || *len || !ms)
        return;
    struct timespec until;
    deadline_after(ms, &until);
    while (!*stop && !*len)
        if (pthread_cond_timedwait(&Q.changed, &Q.mu, &until) == ETIMEDOUT)
            break;
}

void msd_fifo_memory(void **base, size_t *bytes)
{
    pthread_mutex_lock(&Q.mu);
    if (base)
        *base = Q.ready ? Q.samples : NULL;
    if (bytes)
        *bytes = Q.ready ? (size_t)Q.count * Q.size * sizeof *Q.samples : 0;
    pthread_mutex_unlock(&Q.mu);
}

/* A buffer index back onto the unused stack (mutex held).  A buffer that is already there, or that is still
 * queued, is a caller bug (double release / release of a queued buffer): ignored instead of growing the stack
 * past its allocation. */
static void push_unused(unsigned idx)
{
    if (!Q.ready || idx >= Q.count || Q.unused_len >= Q.count)
        return;
    for (unsigned i = 0; i < Q.unused_len; ++i)
        if (Q.unused[i] == idx)
            return;
    for (unsigned i = 0; i < Q.ring_len; ++i)
        if (Q.ring[(Q.ring_first + i) % Q.count] == idx)
            return;
    Q.unused[Q.unused_len++] = idx;
}

/* the record a caller handed back -> its index, or Q.count if it is not one of ours (mutex held) */
static unsigned index_of(const struct msd_mag_buf *buf)
{
    if (!Q.ready || buf < Q.rec || buf >= Q.rec + Q.count)
        return Q.count;
    return (unsigned)(buf - Q.rec);
}

bool msd_fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned overlap)
{
    if (Q.ready || !buffer_count || buffer_size < overlap)
        return false;
    Q.rec = calloc(buffer_count, sizeof *Q.rec);
    Q.samples = calloc((size_t)buffer_count * buffer_size, sizeof *Q.samples);
    Q.ring = calloc(buffer_count, sizeof *Q.ring);
    Q.unused = calloc(buffer_count, sizeof *Q.unused);
    Q.carry = calloc(overlap ? overlap : 1, sizeof *Q.carry);
    if (!Q.rec || !Q.samples || !Q.ring || !Q.unused || !Q.carry) {
        free(Q.rec); free(Q.samples); free(Q.ring); free(Q.unused); free(Q.carry);
        Q.rec = NULL; Q.samples = NULL; Q.ring = NULL; Q.unused = NULL; Q.carry = NULL;
        return false;
    }
    Q.count = buffer_count;
    Q.size = buffer_size;
    Q.overlap = overlap;
    for (unsigned i = 0; i < buffer_count; ++i) {
        Q.rec[i].data = Q.samples + (size_t)i * buffer_size;
        Q.rec[i].totalLength = buffer_size;
        Q.unused[i] = buffer_count - 1 - i; /* buffer 0 is handed out first */
    }
    Q.unused_len = buffer_count;
    Q.ring_first = Q.ring_len = 0;
    Q.carry_valid = false;
    Q.halted = false;
    Q.ready = true;
    return true;
}

void msd_fifo_destroy(void)
{
    if (!Q.ready)
        return;
    free(Q.rec); free(Q.samples); free(Q.ring); free(Q.unused); free(Q.carry);
    Q.rec = NULL; Q.samples = NULL; Q.ring = NULL; Q.unused = NULL; Q.carry = NULL;
    Q.count = Q.ring_len = Q.unused_len = 0;
    Q.ready = false;
}

void msd_fifo_drain(void)
{
    pthread_mutex_lock(&Q.mu);
    while (Q.ready && !Q.halted && Q.ring_len)
        pthread_cond_wait(&Q.changed, &Q.mu);
    pthread_mutex_unlock(&Q.mu);
}

void msd_fifo_halt(void)
{
    pthread_mutex_lock(&Q.mu);
    Q.halted = true;
    while (Q.ready && Q.ring_len) { /* what was queued is unused again */
        const unsigned idx = Q.ring[Q.ring_first];
        Q.ring_first = (Q.ring_first + 1) % Q.count;
        Q.ring_len--;
        push_unused(idx);
    }
    pthread_cond_broadcast(&Q.changed);
    pthread_mutex_unlock(&Q.mu);
}

struct msd_mag_buf *msd_fifo_acquire(uint32_t timeout_ms)
{
    struct msd_mag_buf *b = NULL;
    pthread_mutex_lock(&Q.mu);
    if (Q.ready) {
        wait_for(&Q.halted, &Q.unused_len, timeout_ms);
        if (!Q.halted && Q.unused_len) {
            b = &Q.rec[Q.unused[--Q.unused_len]];
            b->overlap = Q.overlap; /* fifo.h:99-111: a buffer starts out holding only its overlap region */
            b->validLength = Q.overlap;
            b->sampleTimestamp = 0;
            b->sysTimestamp = 0;
            b->flags = (msd_mag_buf_flags)0;
            b->dropped = 0;
            b->next = NULL;
        }
    }
    pthread_mutex_unlock(&Q.mu);
    return b;
}

void msd_fifo_enqueue(struct msd_mag_buf *buf)
{
    if (!buf)
        return;
    pthread_mutex_lock(&Q.mu);
    const unsigned idx = index_of(buf);
    if (idx == Q.count || buf->validLength > buf->totalLength || Q.ring_len >= Q.count) {
        /* not one of this FIFO's buffers (or the FIFO is gone), or filled past its end (fifo.c:172 asserts it): dropped */
        if (idx != Q.count)
            push_unused(idx);
    } else if (Q.halted) { /* fifo.h:92: produced buffers go straight back */
        push_unused(idx);
    } else {
        /* the region in front of the new samples: the previous buffer's tail, or silence at the start
         * of the stream and behind a gap (fifo.h:34-55, MAGBUF_DISCONTINUOUS) */
        if (Q.carry_valid && !(buf->flags & MSD_MAGBUF_DISCONTINUOUS))
            memcpy(buf->data, Q.carry, Q.overlap * sizeof *buf->data);
        else
            memset(buf->data, 0, Q.overlap * sizeof *buf->data);
        if (buf->validLength >= Q.overlap) {
            memcpy(Q.carry, buf->data + (buf->validLength - Q.overlap), Q.overlap * sizeof *buf->data);
            Q.carry_valid = true;
        }
        Q.ring[(Q.ring_first + Q.ring_len) % Q.count] = idx;
        Q.ring_len++;
    }
    pthread_cond_broadcast(&Q.changed);
    pthread_mutex_unlock(&Q.mu);
}

struct msd_mag_buf *msd_fifo_dequeue(uint32_t timeout_ms)
{
    struct msd_mag_buf *b = NULL;
    pthread_mutex_lock(&Q.mu);
    if (Q.ready) {
        wait_for(&Q.halted, &Q.ring_len, timeout_ms);
        if (!Q.halted && Q.ring_len) {
            b = &Q.rec[Q.ring[Q.ring_first]];
            Q.ring_first = (Q.ring_first + 1) % Q.count;
            Q.ring_len--;
            pthread_cond_broadcast(&Q.changed); /* msd_fifo_drain watches the queue length */
        }
    }
    pthread_mutex_unlock(&Q.mu);
    return b;
}

void msd_fifo_release(struct msd_mag_buf *buf)
{
    if (!buf)
        return;
    pthread_mutex_lock(&Q.mu);
    push_unused(index_of(buf));
    pthread_cond_broadcast(&Q.changed);
    pthread_mutex_unlock(&Q.mu);
}
