/*
 * modes_hip_readsb.h -- the reference-shaped host boundary of the MI355X Mode S receive path
 * (libmsd_host.so, plain C): the three interfaces readsb's hot path sits behind, with the
 * reference's own signatures, so that a readsb maintainer binds them by name (included behind the
 * reference's convert.h / fifo.h, the declarations below use the reference's own types):
 *
 *   converter          convert.h:27-45   struct converter_state, iq_convert_fn, init_converter,
 *                                        cleanup_converter
 *   mag_buf FIFO       fifo.h:57-120     struct mag_buf, fifo_create .. fifo_release
 *   "ifile" front-end  sdr.c:41-50,78-98 the five-function sdr_handler (sdr_ifile.h)
 *
 * Everything here is a thin host-C layer over the C-ABI of modes_hip.h; the signal processing runs
 * in the HIP kernels behind it.
 *
 * Binding by the reference's own types is an explicit opt-in of the including file (round 6; it used to follow the
 * reference's include guards CONVERT_H / FIFO_H, which a rename upstream would have flipped silently):
 *
 *   #include "convert.h"
 *   #include "fifo.h"
 *   #define MSD_BIND_REFERENCE_TYPES          (or only MSD_BIND_REFERENCE_CONVERTER / MSD_BIND_REFERENCE_FIFO)
 *   #include "modes_hip_readsb.h"
 *
 * The entry points are then declared with input_format_t / iq_convert_fn / struct mag_buf themselves (a missing
 * reference header is a compile error, not another ABI surface); without the macros this header declares its own,
 * layout-identical msd_* types (tests/c/boundary_ref_layout.c compares them field by field).
 */
#ifndef MODES_HIP_READSB_H
#define MODES_HIP_READSB_H

#include <stdbool.h>
#include <time.h>
#include <stdint.h>

#include "modes_hip.h"

#ifdef MSD_BIND_REFERENCE_TYPES
#ifndef MSD_BIND_REFERENCE_CONVERTER
#define MSD_BIND_REFERENCE_CONVERTER 1
#endif
#ifndef MSD_BIND_REFERENCE_FIFO
#define MSD_BIND_REFERENCE_FIFO 1
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* converter (convert.h:27-45)                                                                */
/* ------------------------------------------------------------------------------------------ */
struct converter_state; /* opaque, as in convert.h:27; here it carries the GPU context */

#ifdef MSD_BIND_REFERENCE_CONVERTER /* the includer has the reference's convert.h in scope and asks for its types */
typedef input_format_t msd_input_format_t;
typedef iq_convert_fn msd_iq_convert_fn;
#else
typedef enum { MSD_INPUT_UC8 = 0, MSD_INPUT_SC16, MSD_INPUT_SC16Q11 } msd_input_format_t; /* convert.h:29-31 */
typedef void (*msd_iq_convert_fn)(void *iq_data, uint16_t *mag_data, unsigned nsamples,
                                  struct converter_state *state, double *out_mean_level,
                                  double *out_mean_power); /* convert.h:33-38 */
#endif

/* init_converter (convert.h:40-43, convert.c:446-491): returns the converter for `format` and a state
 * for it, or NULL (no GPU; unknown format; filter_dc without a sample_rate).  With filter_dc the DC block's constant is
 * exp(-2 pi / sample_rate) for whatever rate the host passes (convert.c:479-482), not only Modes.sample_rate.  The returned
 * function converts on the GPU and writes magnitudes and means bit-identical to convert_uc8_nodc /
 * convert_sc16_nodc / convert_sc16q11_nodc (convert.c:63-111,215-253,332-370) -- convert_sc16q11_table
 * (:264-328) after msd_converter_set_sc16q11_table_bits -- or, with filter_dc, to the convert_*_generic
 * functions (:113-213,374-423), whose DC estimate lives in the state and runs on from call to call like
 * the reference's.  Either out pointer may be NULL (convert.c:104-110).  It is void like the reference's:
 * after a device failure the magnitudes are zero and msd_converter_error(state) says why. */
msd_iq_convert_fn msd_init_converter(msd_input_format_t format, double sample_rate, int filter_dc,
                                     struct converter_state **out_state);
void msd_cleanup_converter(struct converter_state *state); /* convert.h:45 */
/* the GPU device the next msd_init_converter uses (default 0); the context behind a state, for callers
 * that want to demodulate with the same one (msd_demodulate_magbuf); the last error text or "" */
void msd_converter_set_device(int device);
/* the SC16Q11_TABLE_BITS of the build being replaced -- a compile-time constant of the reference, so a property of the
 * host program here too: the one process-wide setting of this layer (with the device number above), read when a converter
 * is made and kept in its state from then on.  init_converter(INPUT_SC16Q11, ..., 0, ...) then hands out the table converter
 * (convert.c:437-438), as the reference's does.  0 (default): the float path.  Returns 0, or -EINVAL for a value outside
 * 0..11 (nothing changes then; msd_ifileOpen and the replay tool report it). */
int msd_converter_set_sc16q11_table_bits(int bits);
msd_ctx *msd_converter_context(struct converter_state *state);
const char *msd_converter_error(const struct converter_state *state);

/* ------------------------------------------------------------------------------------------ */
/* mag_buf FIFO (fifo.h:57-120)                                                               */
/* ------------------------------------------------------------------------------------------ */
#ifdef MSD_BIND_REFERENCE_FIFO /* the includer has the reference's fifo.h in scope: `struct msd_mag_buf` IS its `struct mag_buf`, so that msd_fifo_* and
               * msd_demodulate2400[AC] are assignable to pointers of the reference's own function types without a cast
               * (tests/c/boundary_ref_bind.c; the two layouts are compared field by field in boundary_ref_layout.c) */
#define msd_mag_buf mag_buf
typedef mag_buf_flags msd_mag_buf_flags;
#define MSD_MAGBUF_DISCONTINUOUS MAGBUF_DISCONTINUOUS
#else
typedef enum {
    MSD_MAGBUF_DISCONTINUOUS = 1, /* fifo.h:30-32 */
} msd_mag_buf_flags;

/* struct mag_buf, fifo.h:57-73 (field for field; `next` is kept for the layout and always NULL) */
struct msd_mag_buf {
    uint16_t *data;
    unsigned totalLength;
    unsigned validLength;
    unsigned overlap;
    uint64_t sampleTimestamp;
    uint64_t sysTimestamp;
    msd_mag_buf_flags flags;
    double mean_level;
    double mean_power;
    unsigned dropped;
    struct msd_mag_buf *next;
};
#endif

/* Same calls, same meaning as fifo.h:80-120.  Two defects of fifo.c are not reproduced (SURVEY.md
 * 8(b)): every enqueued buffer is delivered, in order, however deep the queue (fifo.c:192-197 never
 * advances its tail pointer and loses buffers once two are queued), and a timeout really times out
 * (fifo.c:141,219 test pthread_cond_timedwait's result with `< 0`, which never holds). */
bool msd_fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned overlap); /* fifo.h:80 */
void msd_fifo_destroy(void);                                                         /* fifo.h:84 */
void msd_fifo_drain(void);                                                           /* fifo.h:87 */
void msd_fifo_halt(void);                                                            /* fifo.h:94 */
struct msd_mag_buf *msd_fifo_acquire(uint32_t timeout_ms);                           /* fifo.h:99 */
void msd_fifo_enqueue(struct msd_mag_buf *buf);                                      /* fifo.h:111 */
struct msd_mag_buf *msd_fifo_dequeue(uint32_t timeout_ms);                           /* fifo.h:117 */
void msd_fifo_release(struct msd_mag_buf *buf);                                      /* fifo.h:120 */
/* (not in fifo.h) where all the buffers' samples lie, for a host that wants to page-lock them (msd_host_register) */
void msd_fifo_memory(void **base, size_t *bytes);

/* ------------------------------------------------------------------------------------------ */
/* "ifile" SDR front-end (sdr.c:41-50,78-98; sdr_ifile.c)                                     */
/* ------------------------------------------------------------------------------------------ */
/* demodulators (demod_2400.h:37-38)                                                          */
/* ------------------------------------------------------------------------------------------ */
/* `void demodulate2400(struct mag_buf *)` / `void demodulate2400AC(struct mag_buf *)` for the consumer loop of
 * readsb.c:820-855.  What the reference's functions take from the global `Modes` is bound once: the GPU context and
 * the message sink that stands for useModesMessage().  The demodulator needs a context OF ITS OWN, made with msd_create
 * from the receiver's options (threshold, nfix_crc, mode_ac; as msd_sdr_ifile.c does) -- not the converter's
 * (msd_converter_context() has fixed options, and a context has no lock: the reader thread in msd_convert and the
 * consumer in msd_demodulate2400 would race on it).  msd_demodulate2400 runs the buffer through the GPU (Mode S, Mode A/C if the
 * context has it, icaoFilterExpire) and delivers the Mode S messages; msd_demodulate2400AC on the SAME buffer
 * then delivers its Mode A/C replies -- the reference's order.  void like the reference's: after a device
 * failure nothing is delivered and msd_demod_error() says why.  ctx == NULL unbinds. */
int msd_demod_bind(msd_ctx *ctx, int mode_ac, msd_message_fn sink, void *user);
void msd_demodulate2400(struct msd_mag_buf *mag);   /* demod_2400.h:37 */
void msd_demodulate2400AC(struct msd_mag_buf *mag); /* demod_2400.h:38 */
const char *msd_demod_error(void);

/* ------------------------------------------------------------------------------------------ */
/* Shaped like the reference's handler so it can be registered in sdr_handlers[]:
 *     { msd_ifileInitConfig, msd_ifileHandleOption, msd_ifileOpen, msd_ifileRun, msd_ifileClose,
 *       "ifile", SDR_IFILE, 0 }
 * Two run modes:
 *   MSD_IFILE_MAGBUF  the literal drop-in: blocks of 131072 samples are converted on the GPU (the
 *                     msd_init_converter function), pushed through the mag_buf FIFO, and the consumer
 *                     calls the demodulate2400-shaped msd_demodulate_magbuf per buffer;
 *   MSD_IFILE_FUSED   the fast path: many blocks per call go straight to msd_launch_host, which runs
 *                     the fused convert+demodulate kernel; magnitudes never leave the GPU.
 * Both deliver the same ordered messages. */
enum { MSD_OPT_IFILE_NAME = 1, MSD_OPT_IFILE_FORMAT, MSD_OPT_IFILE_THROTTLE, MSD_OPT_IFILE_MODE };
enum { MSD_IFILE_FUSED = 0, MSD_IFILE_MAGBUF = 1 };

/* receiver options the handler cannot see from its own options (Modes.* in the reference) */
typedef struct msd_receiver_options {
    int preamble_threshold; /* Modes.preambleThreshold */
    int nfix_crc;           /* Modes.nfix_crc */
    int mode_ac;            /* Modes.mode_ac */
    int device;
    unsigned batch_buffers; /* fused mode: buffers per GPU batch (default 64) */
    msd_message_fn sink;    /* useModesMessage */
    void *sink_user;
    int dc_filter;          /* Modes.dc_filter (--dcfilter, readsb.c:486) */
    int sc16q11_table_bits; /* the SC16Q11_TABLE_BITS the host was built with (convert.c:264-328; 0: not defined) */
} msd_receiver_options;

/* What the reference's handler reaches through the global `Modes` and sdr.h (sdr_ifile.c:86,178-184,236):
 * any of them may be NULL. */
typedef struct msd_ifile_hooks {
    int (*should_exit)(void);    /* `Modes.exit`: polled between blocks, ends msd_ifileRun */
    void (*monitor)(void);       /* sdrMonitor(), once per block (sdr_ifile.c:184) */
    void (*at_eof)(void);        /* `Modes.exit = 1` once the last block has been delivered (sdr_ifile.c:236) */
    void (*device_selected)(void); /* `Modes.sdr_type = SDR_IFILE` when the file name option arrives (sdr_ifile.c:86) */
} msd_ifile_hooks;

/* --throttle pacing (sdr_ifile.c:168-169,218-226): msd_pacer_wait blocks until the buffer may be released, then
 * moves the deadline on by samples / sample_rate seconds.  Used by msd_ifileRun when the throttle option is set (one
 * buffer per batch then, collected at once); exposed for hosts that feed msd_launch_host themselves. */
typedef struct msd_pacer {
    struct timespec next;
    double sample_rate;
} msd_pacer;
void msd_pacer_start(msd_pacer *p, double sample_rate);
void msd_pacer_wait(msd_pacer *p, uint64_t samples);

void msd_ifileInitConfig(void);                    /* sdr_ifile.c:70-80 */
/* sdr_ifile.c:82-107.  `key` is compared with the values registered through msd_ifileSetOptionKeys --
 * the reference's OptIfileName / OptIfileFormat / OptIfileThrottle (readsb.h:615-617; enum values only
 * the reference's own build knows) -- or, before any registration, with MSD_OPT_IFILE_*.  Unknown keys
 * are accepted and ignored, as the reference's handler does. */
bool msd_ifileHandleOption(int key, char *arg);
bool msd_ifileOpen(void);                          /* sdr_ifile.c:115-162 */
void msd_ifileRun(void);                           /* sdr_ifile.c:164-237 (blocks until EOF or should_exit) */
void msd_ifileClose(void);                         /* sdr_ifile.c:239-255 */
/* registrations survive msd_ifileInitConfig (they describe the host program, not a run); mode_key < 0:
 * no option selects the run mode */
void msd_ifileSetOptionKeys(int name_key, int format_key, int throttle_key, int mode_key);
void msd_ifileSetHooks(const msd_ifile_hooks *hooks);
void msd_ifileSetReceiver(const msd_receiver_options *opt);
int msd_ifileGetStats(msd_stats *st);
/* What the last msd_ifileRun cost, on the host's clock (CLOCK_MONOTONIC).  "Buffer" = one 131072-sample block of the
 * reader loop (sdr_ifile.c:192-216).  MSD_IFILE_MAGBUF: convert_us is the iq_convert_fn call (upload, conversion,
 * magnitudes and means back), demod_us the demodulate2400-shaped call on the consumer thread (readsb.c:846-851).
 * MSD_IFILE_FUSED with --throttle: demod_us is msd_launch_host + msd_collect of the buffer.  latency_us: from the moment
 * a buffer was released to the demodulator (fifo_enqueue / the pacer's deadline) to its last message handed to the sink;
 * deadline_misses: buffers that were not through by the time the next one was due (one buffer period = 54.6 ms at
 * 2.4 MSPS) -- meaningful under --throttle only.  Percentiles over at most the first 65536 buffers. */
typedef struct msd_ifile_timing {
    uint64_t buffers, samples;
    double wall_s;
    double convert_us_p50, convert_us_p99;
    double demod_us_p50, demod_us_p99, demod_us_max;
    double latency_us_p50, latency_us_p99, latency_us_max;
    uint64_t deadline_misses;
    double reader_wait_s, consumer_wait_s; /* MSD_IFILE_MAGBUF: time the reader spent waiting for a free buffer (fifo_acquire)
                                              and the consumer for a filled one (fifo_dequeue): who is the bottleneck */
} msd_ifile_timing;
int msd_ifileGetTiming(msd_ifile_timing *t);
const char *msd_ifileLastError(void);

#ifdef __cplusplus
}
#endif
#endif
