/*
 * ref_fifo_link.c -- test infrastructure.  Lets the reference's own fifo.c (compiled unmodified from
 * /root/reference, see oracle/Makefile target _ref) load as a shared library: fifo.c calls one function
 * it does not define, get_deadline() of util.c, and util.c cannot be compiled here (it includes
 * readsb.h, which needs the absent <protobuf-c/protobuf-c.h>).  This is that one function as util.h:55
 * declares it and util.c:94-99 describes it: the CLOCK_REALTIME time timeout_ms from now.
 * Nothing in the product links or loads this file.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <time.h>

void get_deadline(uint32_t timeout_ms, struct timespec *ts)
{
    clock_gettime(CLOCK_REALTIME, ts);
    ts->tv_sec += timeout_ms / 1000;
    ts->tv_nsec += (long)(timeout_ms % 1000) * 1000000L;
    while (ts->tv_nsec >= 1000000000L) {
        ts->tv_nsec -= 1000000000L;
        ts->tv_sec++;
    }
}
