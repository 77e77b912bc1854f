/* msd_sdr_ifile.c -- see include/modes_hip_readsb.h.  Host C, like the reference's sdr_ifile.c; all the
 * signal processing happens behind the C-ABI of modes_hip.h. */
#define _GNU_SOURCE
#include "modes_hip_readsb.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "modes_hip_readsb.h"

static struct {
    char *filename;
    int format; /* MSD_FMT_* */
    int mode;
    bool throttle; /* --throttle: buffers are released at the rate the receiver would deliver them (sdr_ifile.c:218-226) */
    int fd;
    bool regular;  /* a regular file: large turns are read by several threads with pread (read_fully) */
    off_t pos;     /* ... from here */
    /* fused mode: the page-locked turns the reader fills while the batches before them are in flight.  Like the reference's
     * read buffer (sdr_ifile.c:141-148, 240-252) they are the handler's from Open to Close -- page-locking 16 MiB takes 3 ms
     * and releasing it 2: in the run itself five of them were 24 of the 29 ms a 10 s capture took -- and a regular file gets
     * no more of them than it has turns. */
    char *ring[MSD_PIPELINE_DEPTH + 1];
    unsigned nring;
    unsigned bytes_per_sample;
    char *readbuf;
    size_t readbuf_bytes;
    msd_receiver_options rx;
    msd_ctx *ctx;
    char err[256];
    atomic_int exit_flag; /* the reader tells the consumer thread that the queue has been drained (ThreadSanitizer found the plain int it was) */
    /* magbuf mode: the converter of msd_init_converter (convert.h:40-43) and its state, like sdr_ifile.c:58-68 */
    msd_iq_convert_fn converter;
    struct converter_state *converter_state;
    /* buffers handed to the consumer and not yet demodulated */
    pthread_mutex_t mu;
    pthread_cond_t idle;
    int in_flight;
    /* msd_ifileGetTiming: per-buffer clocks of the last run (reader writes release_ns / convert_us of buffer k before it
     * hands the buffer over, the consumer reads them after it took it: ordered by the FIFO's mutex) */
    struct run_clock {
        uint64_t buffers, samples, misses;
        double wall_s;
        uint64_t *release_ns; /* [TIMING_CAP] */
        float *convert_us, *demod_us, *latency_us;
        uint64_t nconv, ndemod;
        uint64_t reader_wait_ns, consumer_wait_ns;
    } T;
} F;

enum { TIMING_CAP = 65536 };
static void warm_up(void);

static uint64_t now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void timing_reset(void)
{
    struct run_clock *T = &F.T;
    if (!T->release_ns) {
        T->release_ns = calloc(TIMING_CAP, sizeof *T->release_ns);
        T->convert_us = calloc(TIMING_CAP, sizeof *T->convert_us);
        T->demod_us = calloc(TIMING_CAP, sizeof *T->demod_us);
        T->latency_us = calloc(TIMING_CAP, sizeof *T->latency_us);
        if (!T->release_ns || !T->convert_us || !T->demod_us || !T->latency_us) {
            /* all four or none: release_ns == NULL is what every user of the arrays tests (the run goes on
             * untimed, msd_ifileGetTiming says -EINVAL) */
            free(T->release_ns);
            free(T->convert_us);
            free(T->demod_us);
            free(T->latency_us);
            T->release_ns = NULL;
            T->convert_us = T->demod_us = T->latency_us = NULL;
        }
    }
    T->buffers = T->samples = T->misses = T->nconv = T->ndemod = 0;
    T->reader_wait_ns = T->consumer_wait_ns = 0;
    T->wall_s = 0;
}

/* buffer k (in delivery order) is through: its demodulate call took `demod_ns`, its messages are with the sink now */
static void timing_done(uint64_t k, uint64_t demod_ns, uint64_t now, uint64_t samples)
{
    struct run_clock *T = &F.T;
    T->buffers++;
    T->samples += samples;
    if (!T->release_ns || k >= TIMING_CAP)
        return;
    const uint64_t lat = now - T->release_ns[k];
    T->demod_us[k] = (float)(demod_ns * 1e-3);
    T->latency_us[k] = (float)(lat * 1e-3);
    T->ndemod = k + 1;
    if ((double)lat > (double)MSD_CHUNK_SAMPLES * 1e9 / 2400000.0)
        T->misses++;
}

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static void percentiles(float *v, uint64_t n, double *p50, double *p99, double *pmax)
{
    *p50 = *p99 = 0;
    if (pmax)
        *pmax = 0;
    if (!n)
        return;
    qsort(v, n, sizeof *v, cmp_float);
    *p50 = v[n / 2];
    *p99 = v[(n * 99) / 100 < n ? (n * 99) / 100 : n - 1];
    if (pmax)
        *pmax = v[n - 1];
}

int msd_ifileGetTiming(msd_ifile_timing *t)
{
    struct run_clock *T = &F.T;
    if (!t || !T->release_ns)
        return -EINVAL;
    memset(t, 0, sizeof *t);
    t->buffers = T->buffers;
    t->samples = T->samples;
    t->wall_s = T->wall_s;
    t->deadline_misses = T->misses;
    t->reader_wait_s = (double)T->reader_wait_ns * 1e-9;
    t->consumer_wait_s = (double)T->consumer_wait_ns * 1e-9;
    percentiles(T->convert_us, T->nconv, &t->convert_us_p50, &t->convert_us_p99, NULL);
    percentiles(T->demod_us, T->ndemod, &t->demod_us_p50, &t->demod_us_p99, &t->demod_us_max);
    percentiles(T->latency_us, T->ndemod, &t->latency_us_p50, &t->latency_us_p99, &t->latency_us_max);
    T->nconv = T->ndemod = 0; /* (sorted in place: one reading per run) */
    return 0;
}

/* what describes the host program rather than a run: survives msd_ifileInitConfig */
static struct {
    int name_key, format_key, throttle_key, mode_key;
    msd_ifile_hooks hooks;
} G = {MSD_OPT_IFILE_NAME, MSD_OPT_IFILE_FORMAT, MSD_OPT_IFILE_THROTTLE, MSD_OPT_IFILE_MODE, {NULL, NULL, NULL, NULL}};

/* ---- --throttle (sdr_ifile.c:168-169,218-226): a buffer may be released when the one before it has "played", i.e.
 * samples / rate seconds after that one was released; the first one at once.  Absolute deadlines on CLOCK_MONOTONIC,
 * so a slow consumer does not make the replay drift. ---- */
void msd_pacer_start(msd_pacer *p, double sample_rate)
{
    clock_gettime(CLOCK_MONOTONIC, &p->next);
    p->sample_rate = sample_rate;
}

void msd_pacer_wait(msd_pacer *p, uint64_t samples)
{
    while (clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &p->next, NULL) == EINTR)
        ;
    /* the time the next buffer can be delivered (normalize_timespec, util.c) */
    const double ns = (double)samples * 1e9 / p->sample_rate;
    const uint64_t total = (uint64_t)p->next.tv_nsec + (uint64_t)ns;
    p->next.tv_sec += (time_t)(total / 1000000000ull);
    p->next.tv_nsec = (long)(total % 1000000000ull);
}

const char *msd_ifileLastError(void)
{
    return F.err;
}

void msd_ifileSetOptionKeys(int name_key, int format_key, int throttle_key, int mode_key)
{
    G.name_key = name_key;
    G.format_key = format_key;
    G.throttle_key = throttle_key;
    G.mode_key = mode_key;
}

void msd_ifileSetHooks(const msd_ifile_hooks *hooks)
{
    if (hooks)
        G.hooks = *hooks;
    else
        memset(&G.hooks, 0, sizeof G.hooks);
}

static bool host_wants_exit(void)
{
    return G.hooks.should_exit && G.hooks.should_exit();
}

void msd_ifileInitConfig(void)
{
    memset(&F, 0, sizeof F);
    F.format = MSD_FMT_UC8;
    F.fd = -1;
    F.rx.preamble_threshold = 58; /* PREAMBLE_THRESHOLD_DEFAULT, demod_2400.h:31 */
    F.rx.nfix_crc = 1;            /* readsb.c:169 */
    F.rx.batch_buffers = 64;
}

void msd_ifileSetReceiver(const msd_receiver_options *opt)
{
    F.rx = *opt;
    if (!F.rx.batch_buffers)
        F.rx.batch_buffers = 64;
}

bool msd_ifileHandleOption(int key, char *arg)
{
    /* the keys are whatever the host program's option table uses (readsb.h:615-617), see
     * msd_ifileSetOptionKeys; like sdr_ifile.c:82-107 anything else is accepted and ignored */
    if (key == G.name_key) {
        free(F.filename);
        F.filename = arg ? strdup(arg) : NULL;
        if (G.hooks.device_selected)
            G.hooks.device_selected(); /* Modes.sdr_type = SDR_IFILE, sdr_ifile.c:86 */
    } else if (key == G.format_key) {
        if (!arg) {
            snprintf(F.err, sizeof F.err, "Input format missing (supported values: UC8, SC16, SC16Q11)");
            return false;
        }
        if (!strcasecmp(arg, "uc8"))
            F.format = MSD_FMT_UC8;
        else if (!strcasecmp(arg, "sc16"))
            F.format = MSD_FMT_SC16;
        else if (!strcasecmp(arg, "sc16q11"))
            F.format = MSD_FMT_SC16Q11;
        else {
            snprintf(F.err, sizeof F.err, "Input format '%s' not understood (supported values: UC8, SC16, SC16Q11)", arg);
            return false;
        }
    } else if (key == G.throttle_key) {
        F.throttle = true;
    } else if (G.mode_key >= 0 && key == G.mode_key) {
        F.mode = (arg && !strcasecmp(arg, "magbuf")) ? MSD_IFILE_MAGBUF : MSD_IFILE_FUSED;
    }
    return true;
}

bool msd_ifileOpen(void)
{
    if (!F.filename) {
        snprintf(F.err, sizeof F.err, "SDR type 'ifile' requires an --ifile argument");
        return false;
    }
    if (!strcmp(F.filename, "-"))
        F.fd = STDIN_FILENO;
    else if ((F.fd = open(F.filename, O_RDONLY)) < 0) {
        snprintf(F.err, sizeof F.err, "ifile: could not open %s: %s", F.filename, strerror(errno));
        return false;
    }
    off_t file_bytes = -1;
    {
        struct stat sb;
        F.regular = F.fd != STDIN_FILENO && fstat(F.fd, &sb) == 0 && S_ISREG(sb.st_mode);
        F.pos = 0;
        if (F.regular)
            file_bytes = sb.st_size;
    }
    F.bytes_per_sample = (F.format == MSD_FMT_UC8) ? 2 : 4;
    const unsigned nbuf = (F.mode == MSD_IFILE_FUSED) ? F.rx.batch_buffers : 1;
    F.readbuf_bytes = (size_t)F.bytes_per_sample * MSD_CHUNK_SAMPLES * nbuf;
    if (posix_memalign((void **)&F.readbuf, 64, F.readbuf_bytes)) {
        snprintf(F.err, sizeof F.err, "ifile: failed to allocate read buffer");
        msd_ifileClose();
        return false;
    }
    if (F.mode == MSD_IFILE_MAGBUF) {
        /* the literal drop-in: the converter comes from init_converter's twin (sdr_ifile.c:150-153) and
         * keeps its own state; the demodulator's context is created below, as for the fused path */
        msd_converter_set_device(F.rx.device);
        if (msd_converter_set_sc16q11_table_bits(F.rx.sc16q11_table_bits)) {
            snprintf(F.err, sizeof F.err, "ifile: SC16Q11 table bits %d outside 0..11", F.rx.sc16q11_table_bits);
            msd_ifileClose();
            return false;
        }
        F.converter = msd_init_converter((msd_input_format_t)(F.format == MSD_FMT_UC8 ? 0 : F.format == MSD_FMT_SC16 ? 1 : 2),
                                         2400000.0, F.rx.dc_filter, &F.converter_state);
        if (!F.converter) {
            snprintf(F.err, sizeof F.err, "ifile: can't initialize sample converter");
            msd_ifileClose();
            return false;
        }
    }
    msd_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = F.rx.device;
    cfg.format = F.format;
    cfg.preamble_threshold = F.rx.preamble_threshold;
    cfg.nfix_crc = F.rx.nfix_crc;
    cfg.mode_ac = F.rx.mode_ac;
    if (F.format == MSD_FMT_SC16Q11 && !F.rx.dc_filter && F.mode == MSD_IFILE_FUSED)
        cfg.sc16q11_table_bits = F.rx.sc16q11_table_bits; /* a host built with -DSC16Q11_TABLE_BITS (convert.c:437-438) */
    if (F.rx.dc_filter && F.mode == MSD_IFILE_FUSED)
        cfg.flags |= MSD_CFG_DC_FILTER; /* init_converter(..., Modes.dc_filter, ...), sdr_ifile.c:150-153 (the literal drop-in's
                                           converter above has it; its demodulator gets magnitudes) */
    cfg.max_batch_samples = (uint64_t)MSD_CHUNK_SAMPLES * (F.mode == MSD_IFILE_MAGBUF ? 12u /* MAGBUF_BATCH */ : nbuf);
    int rc = msd_create(&cfg, &F.ctx);
    if (rc) {
        snprintf(F.err, sizeof F.err, "ifile: msd_create failed: %s", strerror(-rc));
        msd_ifileClose();
        return false;
    }
    if (F.mode == MSD_IFILE_FUSED) {
        const size_t turn = F.throttle ? (size_t)MSD_CHUNK_SAMPLES * F.bytes_per_sample : F.readbuf_bytes;
        unsigned want = MSD_PIPELINE_DEPTH + 1;
        if (file_bytes >= 0 && (uint64_t)file_bytes / turn + 1 < want)
            want = (unsigned)((uint64_t)file_bytes / turn + 1); /* (the last turn is the short read that ends the capture) */
        for (F.nring = 0; F.nring < want; ++F.nring)
            if (msd_host_alloc(F.ctx, turn, (void **)&F.ring[F.nring])) {
                snprintf(F.err, sizeof F.err, "ifile: %s", msd_last_error(F.ctx));
                msd_ifileClose();
                return false;
            }
    }
    warm_up();
    return true;
}

/* The first launch of every kernel loads its code object (8 ms for the first demodulate call, as much again for the
 * converter): one block of silence through the path the run will take, here, where a receiver opens its device -- not in
 * the first buffer's latency.  msd_reset() puts filter, clocks, counters and the DC block's state back to the start. */
static void warm_up(void)
{
    const size_t n = MSD_CHUNK_SAMPLES;
    char *iq = calloc(n, F.bytes_per_sample);
    uint16_t *mag = calloc(n + MSD_OVERLAP, sizeof *mag);
    if (iq && mag) {
        /* receiver noise, not silence: preamble candidates, so that the resolve stage (its thread pool on the host side) has
         * been through a buffer as well */
        uint32_t x = 2463534242u;
        for (size_t i = 0; i < n * F.bytes_per_sample; ++i) {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            iq[i] = F.bytes_per_sample == 2 ? (char)(125 + (x >> 29)) : (char)((i & 1) ? ((x >> 31) ? 0xff : 0) : (x >> 26));
        }
        if (F.mode == MSD_IFILE_MAGBUF) {
            double level = 0, power = 0;
            F.converter(iq, mag + MSD_OVERLAP, (unsigned)n, F.converter_state, &level, &power);
            (void)msd_reset(msd_converter_context(F.converter_state));
            (void)msd_demodulate_magbuf(F.ctx, mag, (unsigned)(n + MSD_OVERLAP), MSD_OVERLAP, 0, 0, level, power, NULL, NULL);
        } else if (msd_launch_host(F.ctx, iq, n, 0) == 0) {
            (void)msd_collect(F.ctx, NULL, NULL);
        }
        (void)msd_reset(F.ctx);
    }
    free(iq);
    free(mag);
}

/* A turn of the fused loop reads 16 MiB.  From the page cache that is a memcpy by the kernel, 5 GB/s on one core -- which was the
 * whole replay's bound (2.7 Gsamples/s for a capture in /dev/shm against 27 from memory over PCIe).  A regular file's large turns
 * are therefore read by READ_THREADS threads with pread, each its slice of the turn; pipes, stdin and single buffers are read in
 * order as before.  A short slice ends the capture exactly as a short read does (sdr_ifile.c:197-209). */
enum { READ_THREADS = 8, READ_PARALLEL_MIN = 4 << 20 };
struct read_slice {
    char *dst;
    size_t want, got;
    off_t off;
};

static void *read_slice_run(void *arg)
{
    struct read_slice *j = arg;
    j->got = 0;
    while (j->got < j->want) {
        ssize_t n = pread(F.fd, j->dst + j->got, j->want - j->got, j->off + (off_t)j->got);
        if (n <= 0)
            break;
        j->got += (size_t)n;
    }
    return NULL;
}

static size_t read_fully(char *dst, size_t want)
{
    size_t got = 0;
    if (F.regular && want >= (size_t)READ_PARALLEL_MIN) {
        struct read_slice job[READ_THREADS];
        pthread_t th[READ_THREADS];
        bool started[READ_THREADS];
        const size_t per = ((want / READ_THREADS) + 4095) & ~(size_t)4095;
        for (int i = 0; i < READ_THREADS; ++i) {
            const size_t lo = (size_t)i * per < want ? (size_t)i * per : want;
            const size_t hi = lo + per < want && i + 1 < READ_THREADS ? lo + per : want;
            job[i].dst = dst + lo;
            job[i].want = hi - lo;
            job[i].off = F.pos + (off_t)lo;
            job[i].got = 0;
            started[i] = i > 0 && job[i].want && pthread_create(&th[i], NULL, read_slice_run, &job[i]) == 0;
        }
        read_slice_run(&job[0]);
        for (int i = 1; i < READ_THREADS; ++i) {
            if (started[i])
                pthread_join(th[i], NULL);
            else if (job[i].want)
                read_slice_run(&job[i]); /* no thread to be had: this one reads the slice itself */
        }
        for (int i = 0; i < READ_THREADS; ++i) { /* the bytes in front of the first short slice */
            got += job[i].got;
            if (job[i].got < job[i].want)
                break;
        }
        F.pos += (off_t)got;
        (void)lseek(F.fd, F.pos, SEEK_SET); /* a later read() goes on from there */
        return got;
    }
    while (got < want) {
        ssize_t n = read(F.fd, dst + got, want - got);
        if (n <= 0)
            break; /* EOF or error: a short read ends the capture (sdr_ifile.c:197-209) */
        got += (size_t)n;
    }
    F.pos += (off_t)got;
    return got;
}

/* the reference's main-thread consumer loop (readsb.c:820-855) for the mag_buf mode.  readsb demodulates one buffer per turn;
 * when this consumer finds several queued (a file replay: the reader is ahead) it hands up to MAGBUF_BATCH consecutive ones
 * to the GPU in one call -- the same messages in the same order, the round trips paid once (msd_demodulate_magbufs).  A
 * paced feed (--throttle, a live receiver) never has more than one waiting, and gets the one-buffer latency. */
enum { MAGBUF_BATCH = 12 };

static void *magbuf_consumer(void *arg)
{
    (void)arg;
    uint64_t k = 0;
    struct msd_mag_buf *held = NULL; /* dequeued, but it has to start a call of its own */
    (void)msd_thread_attach(F.ctx); /* this thread's first HIP call, before the first buffer is there */
    for (;;) {
        struct msd_mag_buf *bufs[MAGBUF_BATCH];
        msd_magbuf_view views[MAGBUF_BATCH];
        unsigned n = 0;
        if (held) {
            bufs[n++] = held;
            held = NULL;
        } else {
            const uint64_t w0 = now_ns();
            struct msd_mag_buf *buf = msd_fifo_dequeue(100);
            F.T.consumer_wait_ns += now_ns() - w0;
            if (!buf) {
                if (atomic_load(&F.exit_flag))
                    break;
                continue;
            }
            bufs[n++] = buf;
        }
        while (n < MAGBUF_BATCH && !F.throttle && bufs[n - 1]->validLength - bufs[n - 1]->overlap == MSD_CHUNK_SAMPLES) {
            struct msd_mag_buf *nb = msd_fifo_dequeue(0); /* only what is there already */
            if (!nb)
                break;
            if (nb->flags & MSD_MAGBUF_DISCONTINUOUS) { /* its overlap region is silence, not the end of the buffer before it */
                held = nb;
                break;
            }
            bufs[n++] = nb;
        }
        for (unsigned i = 0; i < n; ++i) {
            views[i].data = bufs[i]->data;
            views[i].validLength = bufs[i]->validLength;
            views[i].overlap = bufs[i]->overlap;
            views[i].sampleTimestamp = bufs[i]->sampleTimestamp;
            views[i].sysTimestamp = bufs[i]->sysTimestamp;
            views[i].mean_level = bufs[i]->mean_level;
            views[i].mean_power = bufs[i]->mean_power;
        }
        const uint64_t t0 = now_ns();
        int rc = msd_demodulate_magbufs(F.ctx, views, n, F.rx.sink, F.rx.sink_user);
        const uint64_t t1 = now_ns();
        for (unsigned i = 0; i < n; ++i) {
            timing_done(k++, (t1 - t0) / n, t1, bufs[i]->validLength - bufs[i]->overlap);
            msd_fifo_release(bufs[i]);
        }
        pthread_mutex_lock(&F.mu); /* (F.err is written by the reader thread too: under the lock) */
        if (rc)
            snprintf(F.err, sizeof F.err, "demodulate: %s", msd_last_error(F.ctx));
        F.in_flight -= (int)n;
        pthread_cond_signal(&F.idle);
        pthread_mutex_unlock(&F.mu);
    }
    return NULL;
}

static void run_magbuf(void)
{
    if (!msd_fifo_create(12, MSD_CHUNK_SAMPLES + MSD_OVERLAP, MSD_OVERLAP)) { /* readsb.c:200 */
        snprintf(F.err, sizeof F.err, "Out of memory allocating FIFO");
        return;
    }
    /* page-lock what the two threads hand to the GPU: the FIFO's sample arrays and the block buffer (msd_host_register) */
    void *fifo_mem = NULL;
    size_t fifo_bytes = 0;
    msd_fifo_memory(&fifo_mem, &fifo_bytes);
    const bool fifo_pinned = fifo_mem && msd_host_register(F.ctx, fifo_mem, fifo_bytes) == 0;
    const bool read_pinned = msd_host_register(F.ctx, F.readbuf, F.readbuf_bytes) == 0;
    pthread_t consumer;
    atomic_store(&F.exit_flag, 0);
    pthread_mutex_init(&F.mu, NULL);
    pthread_cond_init(&F.idle, NULL);
    F.in_flight = 0;
    pthread_create(&consumer, NULL, magbuf_consumer, NULL);
    uint64_t sample_counter = 0, kbuf = 0;
    bool eof = false;
    msd_pacer pacer;
    msd_pacer_start(&pacer, 2400000.0); /* Modes.sample_rate, readsb.c:195 */
    /* The reference's reader reads a block, converts it, hands it over (sdr_ifile.c:192-216).  Here the conversion runs on
     * the GPU, so the next block is read while it does (msd_convert_begin / _end on the converter's own context; two block
     * buffers): same order, same converter state from block to block, same buffers out.  --throttle keeps the plain order:
     * a paced reader has nothing to gain and would hold a block back. */
    msd_ctx *conv_ctx = msd_converter_context(F.converter_state);
    const size_t want = (size_t)MSD_CHUNK_SAMPLES * F.bytes_per_sample;
    char *rb[2] = {F.readbuf, NULL};
    bool rb1_pinned = false;
    if (!F.throttle && conv_ctx && posix_memalign((void **)&rb[1], 64, want) == 0)
        rb1_pinned = msd_host_register(F.ctx, rb[1], want) == 0;
    const bool overlap_read = rb[1] != NULL;
    size_t got = read_fully(rb[0], want);
    unsigned cur = 0;
    while (!host_wants_exit()) {
        const uint64_t a0 = now_ns();
        struct msd_mag_buf *out = msd_fifo_acquire(100);
        F.T.reader_wait_ns += now_ns() - a0;
        if (!out)
            continue;
        if (G.hooks.monitor)
            G.hooks.monitor(); /* sdrMonitor(), sdr_ifile.c:184 */
        out->sampleTimestamp = (uint64_t)(sample_counter * 12e6 / 2400000.0); /* sdr_ifile.c:187 */
        out->sysTimestamp = out->sampleTimestamp / 12000U;                    /* startup_time = 0 */
        if (got < want)
            eof = true;
        const unsigned samples = (unsigned)(got / F.bytes_per_sample);
        const uint64_t c0 = now_ns();
        size_t got_next = 0;
        if (overlap_read) {
            const bool begun = msd_convert_begin(conv_ctx, rb[cur], &out->data[out->overlap], samples) == 0;
            if (!eof)
                got_next = read_fully(rb[cur ^ 1], want); /* ... while the GPU converts this one */
            if (!begun) {
                F.converter(rb[cur], &out->data[out->overlap], samples, F.converter_state, &out->mean_level, &out->mean_power);
            } else if (msd_convert_end(conv_ctx, &out->mean_level, &out->mean_power)) {
                pthread_mutex_lock(&F.mu);
                snprintf(F.err, sizeof F.err, "%s", msd_last_error(conv_ctx));
                pthread_mutex_unlock(&F.mu);
            }
        } else {
            F.converter(rb[0], &out->data[out->overlap], samples, F.converter_state, &out->mean_level,
                        &out->mean_power); /* sdr_ifile.c:214 */
            if (!eof)
                got_next = read_fully(rb[0], want);
        }
        if (F.T.release_ns && kbuf < TIMING_CAP) {
            F.T.convert_us[kbuf] = (float)((now_ns() - c0) * 1e-3); /* (with the next block's read inside it when they overlap) */
            F.T.nconv = kbuf + 1;
        }
        out->validLength = out->overlap + samples;
        out->flags = 0;
        pthread_mutex_lock(&F.mu);
        if (msd_converter_error(F.converter_state)[0])
            snprintf(F.err, sizeof F.err, "%s", msd_converter_error(F.converter_state));
        F.in_flight++;
        pthread_mutex_unlock(&F.mu);
        if (F.throttle)
            msd_pacer_wait(&pacer, samples); /* sdr_ifile.c:218-226: wait until this buffer may be released */
        if (F.T.release_ns && kbuf < TIMING_CAP)
            F.T.release_ns[kbuf] = now_ns();
        ++kbuf;
        msd_fifo_enqueue(out);
        /* The converter owns a GPU context of its own (msd_init_converter), the consumer demodulates on F.ctx:
         * the next block is read and converted while this one is demodulated, as the reference's reader and
         * main threads overlap (readsb.c:271-285,820-855).  The FIFO is lossless at any depth (msd_fifo.c), so
         * the messages do not depend on how far the reader gets ahead; msd_fifo_acquire() holds it back once
         * all twelve buffers are in use. */
        sample_counter += samples;
        if (eof)
            break;
        got = got_next;
        if (overlap_read)
            cur ^= 1u;
    }
    if (rb[1]) {
        if (rb1_pinned)
            msd_host_unregister(F.ctx, rb[1]);
        free(rb[1]);
    }
    msd_fifo_drain();
    atomic_store(&F.exit_flag, 1);
    msd_fifo_halt(); /* the queue is empty: this only wakes the consumer out of its fifo_dequeue(100 ms) -- round 5's timing showed
                        every replay ending with that timeout, 0.1 s of a 0.13 s run on the 10 s capture */
    pthread_join(consumer, NULL);
    if (read_pinned)
        msd_host_unregister(F.ctx, F.readbuf);
    if (fifo_pinned)
        msd_host_unregister(F.ctx, fifo_mem);
    msd_fifo_destroy();
}

/* The reader thread's loop of sdr_ifile.c:164-237 with the GPU behind it: blocks of `batch_buffers`
 * buffers are read into a ring of page-locked buffers and handed over asynchronously, so the read of
 * the next block, the upload of the previous one and the kernels of the one before overlap. */
static void run_fused(void)
{
    const unsigned RING = F.nring; /* a ring of fewer turns than the pipeline is deep holds the whole file: one turn each */
    char **ring = F.ring;
    if (!RING)
        return;
    {
        bool eof = false;
        int in_flight = 0;
        unsigned k = 0;
        /* --throttle: the real-time form of the same loop.  One buffer per batch, released when the receiver would
         * have delivered it, and its messages collected at once -- what a live feed looks like to the GPU path. */
        const size_t turn_bytes = F.throttle ? (size_t)MSD_CHUNK_SAMPLES * F.bytes_per_sample : F.readbuf_bytes;
        msd_pacer pacer;
        msd_pacer_start(&pacer, 2400000.0);
        while (!eof && !host_wants_exit()) {
            if (G.hooks.monitor)
                G.hooks.monitor(); /* sdrMonitor(), sdr_ifile.c:184 */
            char *buf = ring[k++ % RING]; /* the batch that used it RING turns ago has been collected */
            const size_t got = read_fully(buf, turn_bytes);
            if (got < turn_bytes)
                eof = true;
            const uint64_t samples = got / F.bytes_per_sample;
            int rc = 0;
            if (F.throttle) {
                msd_pacer_wait(&pacer, samples);
                const uint64_t t0 = now_ns();
                if (F.T.release_ns && k - 1 < TIMING_CAP)
                    F.T.release_ns[k - 1] = t0;
                rc = msd_launch_host(F.ctx, buf, samples, eof ? 1 : 0);
                if (!rc)
                    rc = msd_collect(F.ctx, F.rx.sink, F.rx.sink_user);
                const uint64_t t1 = now_ns();
                timing_done(k - 1, t1 - t0, t1, samples);
                if (rc) {
                    snprintf(F.err, sizeof F.err, "submit: %s", msd_last_error(F.ctx));
                    goto out;
                }
                continue;
            }
            if (in_flight > 0 && (in_flight == MSD_PIPELINE_DEPTH || (unsigned)in_flight + 1 >= RING)) { /* the next turn's buffer must be free by then */
                rc = msd_collect(F.ctx, F.rx.sink, F.rx.sink_user);
                in_flight--;
            }
            if (!rc) {
                rc = msd_launch_host(F.ctx, buf, samples, eof ? 1 : 0);
                in_flight++;
            }
            if (rc) {
                snprintf(F.err, sizeof F.err, "submit: %s", msd_last_error(F.ctx));
                goto out;
            }
        }
        while (in_flight-- > 0)
            if (msd_collect(F.ctx, F.rx.sink, F.rx.sink_user)) {
                snprintf(F.err, sizeof F.err, "collect: %s", msd_last_error(F.ctx));
                break;
            }
    }
out:
    return;
}

void msd_ifileRun(void)
{
    if (F.fd < 0 || !F.ctx)
        return;
    timing_reset();
    const uint64_t w0 = now_ns();
    if (F.mode == MSD_IFILE_MAGBUF)
        run_magbuf();
    else
        run_fused();
    F.T.wall_s = (double)(now_ns() - w0) * 1e-9;
    if (!F.T.samples) { /* the unthrottled fused loop does not clock single buffers */
        msd_stats st;
        if (msd_get_stats(F.ctx, &st) == 0) {
            F.T.buffers = st.buffers;
            F.T.samples = st.samples_processed;
        }
    }
    if (G.hooks.at_eof)
        G.hooks.at_eof(); /* Modes.exit = 1, sdr_ifile.c:236 */
}

int msd_ifileGetStats(msd_stats *st)
{
    return F.ctx ? msd_get_stats(F.ctx, st) : -EINVAL;
}

void msd_ifileClose(void)
{
    if (F.converter) { /* sdr_ifile.c:240-244 */
        msd_cleanup_converter(F.converter_state);
        F.converter = NULL;
        F.converter_state = NULL;
    }
    if (F.ctx) {
        for (unsigned i = 0; i < F.nring; ++i)
            msd_host_free(F.ctx, F.ring[i]);
        F.nring = 0;
        msd_destroy(F.ctx);
        F.ctx = NULL;
    }
    free(F.readbuf);
    F.readbuf = NULL;
    if (F.fd >= 0 && F.fd != STDIN_FILENO)
        close(F.fd);
    F.fd = -1;
    free(F.filename);
    F.filename = NULL;
}
