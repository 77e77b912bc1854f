/* Two-thread (plus a halting third) stress of the mag_buf FIFO for the sanitizer builds (scripts/sanitize.sh): the
 * producer runs as far ahead as the pool of twelve buffers lets it (readsb.c:249: MODES_MAG_BUFFERS 12), the consumer
 * checks order, overlap and content of what it dequeues, and in every second round a third thread halts the queue in
 * mid-stream (fifo.h:89-94): both loops must come home, nothing may be touched after the halt. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "modes_hip_readsb.h"

enum { NBUF = 12, OVERLAP = 326, NEW = 2048, ROUNDS = 6, PER_ROUND = 3000 };
static atomic_int halt_after; /* buffers the consumer sees before the halter strikes; 0: never; -1: struck */
static int produced, consumed, failed;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;

static void nap(long ns)
{
    struct timespec ts = {0, ns};
    nanosleep(&ts, NULL);
}

static void *producer(void *arg)
{
    (void)arg;
    for (int k = 0; k < PER_ROUND; ++k) {
        struct msd_mag_buf *b = msd_fifo_acquire(200);
        if (!b)
            break; /* halted (or the consumer is gone) */
        b->validLength = OVERLAP + NEW;
        for (unsigned i = 0; i < NEW; ++i)
            b->data[OVERLAP + i] = (uint16_t)(k * 7 + i);
        b->sampleTimestamp = (uint64_t)k * NEW * 5;
        b->flags = 0;
        b->dropped = 0;
        msd_fifo_enqueue(b);
        pthread_mutex_lock(&mu);
        ++produced;
        pthread_mutex_unlock(&mu);
    }
    return NULL;
}

static void *consumer(void *arg)
{
    (void)arg;
    int k = 0;
    for (;;) {
        struct msd_mag_buf *b = msd_fifo_dequeue(100);
        if (!b) {
            pthread_mutex_lock(&mu);
            const int done = produced == PER_ROUND && consumed == produced;
            pthread_mutex_unlock(&mu);
            if (done || halt_after < 0)
                break;
            continue;
        }
        int bad = b->validLength != OVERLAP + NEW || b->sampleTimestamp != (uint64_t)k * NEW * 5;
        for (unsigned i = 0; i < NEW && !bad; i += 37)
            bad = b->data[OVERLAP + i] != (uint16_t)(k * 7 + i);
        if (k > 0) /* the overlap region holds the end of the previous buffer (fifo.h:34-55) */
            for (unsigned i = 0; i < OVERLAP && !bad; i += 13)
                bad = b->data[i] != (uint16_t)((k - 1) * 7 + NEW - OVERLAP + i);
        msd_fifo_release(b);
        pthread_mutex_lock(&mu);
        failed += bad;
        ++consumed;
        pthread_mutex_unlock(&mu);
        ++k;
        if (k % 64 == 0)
            nap(200000); /* let the producer run twelve ahead now and then */
    }
    return NULL;
}

static void *halter(void *arg)
{
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&mu);
        const int c = consumed;
        pthread_mutex_unlock(&mu);
        if (c >= halt_after)
            break;
        nap(50000);
    }
    msd_fifo_halt();
    halt_after = -1;
    return NULL;
}

int main(void)
{
    for (int round = 0; round < ROUNDS; ++round) {
        produced = consumed = 0;
        halt_after = (round & 1) ? 500 + 211 * round : 0;
        if (!msd_fifo_create(NBUF, OVERLAP + NEW, OVERLAP))
            return 2;
        pthread_t p, c, h;
        pthread_create(&c, NULL, consumer, NULL);
        pthread_create(&p, NULL, producer, NULL);
        const int with_halt = halt_after > 0;
        if (with_halt)
            pthread_create(&h, NULL, halter, NULL);
        pthread_join(p, NULL);
        if (with_halt)
            pthread_join(h, NULL);
        pthread_join(c, NULL);
        if (with_halt) {
            if (msd_fifo_acquire(0) || msd_fifo_dequeue(0)) /* after a halt both return NULL at once */
                return 3;
        } else {
            if (consumed != PER_ROUND) {
                fprintf(stderr, "round %d: %d of %d buffers arrived\n", round, consumed, PER_ROUND);
                return 4;
            }
            msd_fifo_drain();
        }
        msd_fifo_destroy();
        printf("round %d: produced %d consumed %d%s\n", round, produced, consumed, with_halt ? " (halted in mid-stream)" : "");
    }
    if (failed) {
        fprintf(stderr, "%d buffers arrived out of order or with wrong content\n", failed);
        return 5;
    }
    printf("fifo stress ok\n");
    return 0;
}
