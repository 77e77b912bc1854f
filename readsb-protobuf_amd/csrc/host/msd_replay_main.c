/*
 * msd_replay -- `readsb --device-type ifile --ifile F --iformat X [--fix|--no-fix|--aggressive]
 * [--preamble-threshold N] [--modeac] --raw --quiet-ish` for the part of readsb this repository
 * implements: replays a capture through the GPU receive path and prints one `*hex;` line per
 * accepted message like displayModesMessage does in --raw mode (mode_s.c:1786-1798), or
 * `@<12 hex digit timestamp>hex;` with --mlat.  --net-raw prints the lines of the raw TCP output
 * instead (net_io.c:870-896, upper-case hex), --beast writes Beast binary frames (net_io.c:769-835); both
 * follow modesQueueOutput's forwarding rule (net_io.c:1263-1290: two-bit repairs only with --net-verbatim,
 * which also sends the bytes as received).
 * Counters go to stderr with --stats.
 */
#define _GNU_SOURCE
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "modes_hip_readsb.h"
#include "msd_wire.h"

static int g_mlat, g_net_verbatim; /* Modes.mlat, Modes.net_verbatim */
static uint64_t g_count;

static void print_raw(const msd_message *mm, void *user)
{
    FILE *out = user;
    if (g_mlat && mm->timestampMsg)
        fprintf(out, "@%012" PRIX64, mm->timestampMsg);
    else
        fputc('*', out);
    for (int j = 0; j < mm->msgbits / 8; j++)
        fprintf(out, "%02x", mm->msg[j]);
    fputs(";\n", out);
    g_count++;
}

static void print_net_raw(const msd_message *mm, void *user)
{
    char line[MSD_AVR_MAX];
    /* modesQueueOutput (net_io.c:1263-1290): a message that needed two repairs only with --net-verbatim, and then -- like
     * every message -- with the bytes as received (net_io.c:874) */
    const size_t n = msd_avr_line_out(mm, g_mlat, g_net_verbatim, line);
    if (!n)
        return;
    fwrite(line, 1, n, (FILE *)user);
    g_count++;
}

static void count_only(const msd_message *mm, void *user)
{
    (void)mm;
    (void)user;
    g_count++;
}

static void write_beast(const msd_message *mm, void *user)
{
    uint8_t frame[MSD_BEAST_MAX];
    const size_t n = msd_beast_frame_out(mm, g_net_verbatim, frame); /* net_io.c:1278-1285, :775 */
    if (!n)
        return;
    fwrite(frame, 1, n, (FILE *)user);
    g_count++;
}

/* This tool plays readsb's part towards the handler: its own option keys (as readsb.h:615-617 are readsb's)
 * and the hooks that stand for Modes.exit / sdrMonitor() (sdr_ifile.c:178-184,236). */
enum { OptIfileName = 615, OptIfileFormat, OptIfileThrottle, OptIfilePath };
static volatile int g_exit; /* Modes.exit */
static int host_should_exit(void) { return g_exit; }
static void host_at_eof(void) { g_exit = 1; }

int main(int argc, char **argv)
{
    msd_receiver_options rx;
    memset(&rx, 0, sizeof rx);
    rx.preamble_threshold = 58;
    rx.nfix_crc = 1;
    rx.batch_buffers = 64;
    rx.sink = print_raw;
    rx.sink_user = stdout;
    int want_stats = 0, want_timing = 0;

    const msd_ifile_hooks hooks = {host_should_exit, NULL, host_at_eof, NULL};
    msd_ifileSetOptionKeys(OptIfileName, OptIfileFormat, OptIfileThrottle, OptIfilePath);
    msd_ifileSetHooks(&hooks);
    msd_ifileInitConfig();
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        char *next = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--ifile") && next) { msd_ifileHandleOption(OptIfileName, next); ++i; }
        else if (!strcmp(a, "--iformat") && next) {
            if (!msd_ifileHandleOption(OptIfileFormat, next)) { fprintf(stderr, "%s\n", msd_ifileLastError()); return 1; }
            ++i;
        }
        else if (!strcmp(a, "--throttle")) msd_ifileHandleOption(OptIfileThrottle, NULL);
        else if (!strcmp(a, "--path") && next) { msd_ifileHandleOption(OptIfilePath, next); ++i; }
        else if (!strcmp(a, "--fix")) rx.nfix_crc = 1;
        else if (!strcmp(a, "--no-fix")) rx.nfix_crc = 0;
        else if (!strcmp(a, "--aggressive")) rx.nfix_crc = 2; /* readsb.c:542 */
        else if (!strcmp(a, "--modeac")) rx.mode_ac = 1;
        else if (!strcmp(a, "--dcfilter")) rx.dc_filter = 1; /* readsb.c:486 */
        else if (!strcmp(a, "--mlat")) g_mlat = 1;
        else if (!strcmp(a, "--net-verbatim")) g_net_verbatim = 1; /* readsb.c: Modes.net_verbatim */
        else if (!strcmp(a, "--stats")) want_stats = 1;
        else if (!strcmp(a, "--timing")) want_timing = 1; /* one JSON line on stderr: what the run cost (msd_ifileGetTiming) */
        else if (!strcmp(a, "--no-output")) rx.sink = count_only;
        else if (!strcmp(a, "--net-raw")) rx.sink = print_net_raw;
        else if (!strcmp(a, "--beast")) rx.sink = write_beast;
        else if (!strcmp(a, "--raw") || !strcmp(a, "--quiet")) { /* this tool only has the raw dump */ }
        else if (!strcmp(a, "--device-type") && next) { ++i; /* always ifile */ }
        else if (!strcmp(a, "--device") && next) { rx.device = atoi(next); ++i; }
        else if (!strcmp(a, "--batch-buffers") && next) { rx.batch_buffers = (unsigned)atoi(next); ++i; }
        else if (!strcmp(a, "--sc16q11-table-bits") && next) { rx.sc16q11_table_bits = atoi(next); ++i; } /* a -DSC16Q11_TABLE_BITS=n build */
        else if (!strcmp(a, "--preamble-threshold") && next) {
            long v = strtol(next, NULL, 10); /* readsb.c:503-505 clamps to 40..400 */
            rx.preamble_threshold = (int)(v < 40 ? 40 : (v > 400 ? 400 : v));
            ++i;
        } else {
            fprintf(stderr, "usage: msd_replay --ifile F [--iformat uc8|sc16|sc16q11] [--fix|--no-fix|--aggressive] [--dcfilter] "
                            "[--preamble-threshold N] [--modeac] [--mlat] [--net-raw|--beast|--no-output] [--net-verbatim] [--stats] [--timing] [--throttle] [--path fused|magbuf] "
                            "[--device N] [--sc16q11-table-bits N]\n");
            return 2;
        }
    }
    msd_ifileSetReceiver(&rx);
    if (!msd_ifileOpen()) {
        fprintf(stderr, "%s\n", msd_ifileLastError());
        return 1;
    }
    struct timespec run0, run1;
    clock_gettime(CLOCK_MONOTONIC, &run0);
    msd_ifileRun();
    clock_gettime(CLOCK_MONOTONIC, &run1);
    if (want_stats) /* how long the reader ran: in signal time under --throttle (sdr_ifile.c:218-226) */
        fprintf(stderr, "run_seconds %.3f\n", (double)(run1.tv_sec - run0.tv_sec) + 1e-9 * (double)(run1.tv_nsec - run0.tv_nsec));
    if (want_timing) {
        msd_ifile_timing t;
        if (msd_ifileGetTiming(&t) == 0)
            fprintf(stderr, "{\"buffers\": %" PRIu64 ", \"samples\": %" PRIu64 ", \"messages\": %" PRIu64 ", \"wall_s\": %.6f, \"msamples_per_s\": %.1f, "
                            "\"convert_us_p50\": %.1f, \"convert_us_p99\": %.1f, \"demod_us_p50\": %.1f, \"demod_us_p99\": %.1f, \"demod_us_max\": %.1f, "
                            "\"latency_us_p50\": %.1f, \"latency_us_p99\": %.1f, \"latency_us_max\": %.1f, \"deadline_misses\": %" PRIu64 ", "
                            "\"reader_wait_s\": %.6f, \"consumer_wait_s\": %.6f}\n",
                    t.buffers, t.samples, g_count, t.wall_s, t.wall_s > 0 ? (double)t.samples / t.wall_s * 1e-6 : 0.0, t.convert_us_p50, t.convert_us_p99,
                    t.demod_us_p50, t.demod_us_p99, t.demod_us_max, t.latency_us_p50, t.latency_us_p99, t.latency_us_max, t.deadline_misses,
                    t.reader_wait_s, t.consumer_wait_s);
    }
    if (msd_ifileLastError()[0])
        fprintf(stderr, "%s\n", msd_ifileLastError());
    if (!g_exit) { /* readsb.c:279-281: a reader that returns without the exit flag set is an abnormal exit */
        fprintf(stderr, "reader returned without signalling the end of the capture\n");
        return 2;
    }
    if (want_stats) {
        msd_stats st;
        if (msd_ifileGetStats(&st) == 0) {
            fprintf(stderr, "messages %" PRIu64 "\npreambles %" PRIu64 "\nrejected_bad %" PRIu64
                            "\nrejected_unknown_icao %" PRIu64 "\naccepted %" PRIu64 " %" PRIu64 "\nmodeac %" PRIu64
                            "\nbuffers %" PRIu64 "\n",
                    g_count, st.demod_preambles, st.demod_rejected_bad, st.demod_rejected_unknown_icao,
                    st.demod_accepted[0], st.demod_accepted[1], st.demod_modeac, st.buffers);
        }
    }
    msd_ifileClose();
    return 0;
}
