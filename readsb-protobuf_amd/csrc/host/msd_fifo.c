/* msd_fifo.c -- see msd_fifo.h.  A bounded producer/consumer queue of preallocated magnitude
 * buffers with the reference's overlap rule (fifo.c:179-188): each enqueued buffer is prefixed with
 * the last `overlap` samples of the previous one (zeros for the first or a discontinuous one). */
#define _GNU_SOURCE
#include "msd_fifo.h"

#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static struct {
    pthread_mutex_t mu;
    pthread_cond_t not_empty, empty, have_free;
    struct msd_mag_buf *head, *tail, *freelist;
    bool halted;
    unsigned overlap;
    uint16_t *carry;
} Q = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
       NULL, NULL, NULL, false, 0, NULL};

static void deadline_after(uint32_t timeout_ms, struct timespec *ts)
{
    clock_gettime(CLOCK_REALTIME, ts);
    ts->tv_sec += timeout_ms / 1000;
    ts->tv_nsec += (long)(timeout_ms % 1000) * 1000000L;
    if (ts->tv_nsec >= 1000000000L) {
        ts->tv_sec += 1;
        ts->tv_nsec -= 1000000000L;
    }
}

static void free_list(struct msd_mag_buf *b)
{
    while (b) {
        struct msd_mag_buf *n = b->next;
        free(b->data);
        free(b);
        b = n;
    }
}

void msd_fifo_destroy(void)
{
    free_list(Q.head);
    free_list(Q.freelist);
    Q.head = Q.tail = Q.freelist = NULL;
    free(Q.carry);
    Q.carry = NULL;
    Q.halted = false;
}

bool msd_fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned overlap)
{
    Q.carry = calloc(overlap ? overlap : 1, sizeof Q.carry[0]);
    if (!Q.carry)
        return false;
    Q.overlap = overlap;
    for (unsigned i = 0; i < buffer_count; ++i) {
        struct msd_mag_buf *b = calloc(1, sizeof *b);
        if (b)
            b->data = calloc(buffer_size, sizeof b->data[0]);
        if (!b || !b->data) {
            free(b);
            msd_fifo_destroy();
            return false;
        }
        b->totalLength = buffer_size;
        b->next = Q.freelist;
        Q.freelist = b;
    }
    return true;
}

void msd_fifo_drain(void)
{
    pthread_mutex_lock(&Q.mu);
    while (Q.head && !Q.halted)
        pthread_cond_wait(&Q.empty, &Q.mu);
    pthread_mutex_unlock(&Q.mu);
}

void msd_fifo_halt(void)
{
    pthread_mutex_lock(&Q.mu);
    while (Q.head) {
        struct msd_mag_buf *b = Q.head;
        Q.head = b->next;
        b->next = Q.freelist;
        Q.freelist = b;
    }
    Q.tail = NULL;
    Q.halted = true;
    pthread_cond_broadcast(&Q.not_empty);
    pthread_cond_broadcast(&Q.empty);
    pthread_cond_broadcast(&Q.have_free);
    pthread_mutex_unlock(&Q.mu);
}

struct msd_mag_buf *msd_fifo_acquire(uint32_t timeout_ms)
{
    struct timespec until;
    if (timeout_ms)
        deadline_after(timeout_ms, &until);
    struct msd_mag_buf *b = NULL;
    pthread_mutex_lock(&Q.mu);
    while (!Q.halted && !Q.freelist) {
        if (!timeout_ms || pthread_cond_timedwait(&Q.have_free, &Q.mu, &until) == ETIMEDOUT)
            break;
    }
    if (!Q.halted && Q.freelist) {
        b = Q.freelist;
        Q.freelist = b->next;
        b->overlap = Q.overlap; /* fifo.c:152-158 */
        b->validLength = Q.overlap;
        b->sampleTimestamp = 0;
        b->sysTimestamp = 0;
        b->flags = 0;
        b->next = NULL;
    }
    pthread_mutex_unlock(&Q.mu);
    return b;
}

void msd_fifo_enqueue(struct msd_mag_buf *b)
{
    pthread_mutex_lock(&Q.mu);
    if (Q.halted) {
        b->next = Q.freelist;
        Q.freelist = b;
        pthread_mutex_unlock(&Q.mu);
        return;
    }
    const size_t bytes = Q.overlap * sizeof b->data[0];
    if (b->flags & MSD_MAGBUF_DISCONTINUOUS)
        memset(b->data, 0, bytes);
    else
        memcpy(b->data, Q.carry, bytes);
    memcpy(Q.carry, &b->data[b->validLength - Q.overlap], bytes);
    b->next = NULL;
    if (!Q.head) {
        Q.head = Q.tail = b;
        pthread_cond_signal(&Q.not_empty);
    } else {
        Q.tail->next = b;
        Q.tail = b; /* the line fifo.c:192-197 is missing */
    }
    pthread_mutex_unlock(&Q.mu);
}

struct msd_mag_buf *msd_fifo_dequeue(uint32_t timeout_ms)
{
    struct timespec until;
    if (timeout_ms)
        deadline_after(timeout_ms, &until);
    struct msd_mag_buf *b = NULL;
    pthread_mutex_lock(&Q.mu);
    while (!Q.head && !Q.halted) {
        if (!timeout_ms || pthread_cond_timedwait(&Q.not_empty, &Q.mu, &until) == ETIMEDOUT)
            break;
    }
    if (!Q.halted && Q.head) {
        b = Q.head;
        Q.head = b->next;
        b->next = NULL;
        if (!Q.head) {
            Q.tail = NULL;
            pthread_cond_broadcast(&Q.empty);
        }
    }
    pthread_mutex_unlock(&Q.mu);
    return b;
}

void msd_fifo_release(struct msd_mag_buf *b)
{
    pthread_mutex_lock(&Q.mu);
    if (!Q.freelist)
        pthread_cond_signal(&Q.have_free);
    b->next = Q.freelist;
    Q.freelist = b;
    pthread_mutex_unlock(&Q.mu);
}
