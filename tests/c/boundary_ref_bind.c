/* The binding a readsb maintainer writes, type-checked against the reference's own header text (-I/root/reference,
 * build container only): the reference's convert.h / fifo.h / demod_2400.h first, then MSD_BIND_REFERENCE_TYPES and
 * modes_hip_readsb.h behind them -- which then declares its entry points with the reference's types -- and every replacement assigned, WITHOUT A CAST and
 * under -Wall -Wextra -Werror, to a pointer whose type is taken from the reference's declaration itself
 * (__typeof__(&init_converter) and so on). */
#include <stdint.h> /* convert.h relies on its includer for uint16_t (readsb.h:61 in the reference) */
#include <stdio.h>

#include "convert.h"
#include "fifo.h"
#include "demod_2400.h"
#define MSD_BIND_REFERENCE_TYPES /* the explicit opt-in: the entry points below are declared with the reference's types */
#include "modes_hip_readsb.h"

/* convert.h:27-45 */
static __typeof__(&init_converter) const p_init_converter = msd_init_converter;
static __typeof__(&cleanup_converter) const p_cleanup_converter = msd_cleanup_converter;
/* demod_2400.h:37-38 */
static __typeof__(&demodulate2400) const p_demodulate2400 = msd_demodulate2400;
static __typeof__(&demodulate2400AC) const p_demodulate2400AC = msd_demodulate2400AC;
/* fifo.h:80-120 */
static __typeof__(&fifo_create) const p_fifo_create = msd_fifo_create;
static __typeof__(&fifo_destroy) const p_fifo_destroy = msd_fifo_destroy;
static __typeof__(&fifo_drain) const p_fifo_drain = msd_fifo_drain;
static __typeof__(&fifo_halt) const p_fifo_halt = msd_fifo_halt;
static __typeof__(&fifo_acquire) const p_fifo_acquire = msd_fifo_acquire;
static __typeof__(&fifo_enqueue) const p_fifo_enqueue = msd_fifo_enqueue;
static __typeof__(&fifo_dequeue) const p_fifo_dequeue = msd_fifo_dequeue;
static __typeof__(&fifo_release) const p_fifo_release = msd_fifo_release;

int main(void)
{
    /* the FIFO through the reference-typed pointers, with the reference's struct (readsb.c:820-855 in miniature) */
    if (!p_fifo_create(3, 4096 + 326, 326))
        return 1;
    struct mag_buf *b = p_fifo_acquire(0);
    if (!b || b->totalLength != 4096 + 326 || b->overlap != 326)
        return 2;
    b->validLength = b->totalLength;
    b->flags = MAGBUF_DISCONTINUOUS;
    b->sampleTimestamp = 12345;
    p_fifo_enqueue(b);
    struct mag_buf *c = p_fifo_dequeue(0);
    if (c != b || c->sampleTimestamp != 12345 || !(c->flags & MAGBUF_DISCONTINUOUS))
        return 3;
    p_fifo_release(c);
    p_fifo_halt();
    if (p_fifo_acquire(0))
        return 4;
    p_fifo_drain();
    p_fifo_destroy();
    /* the converter factory with the reference's argument types: NULL without a GPU, a converter with one */
    struct converter_state *st = NULL;
    iq_convert_fn fn = p_init_converter(INPUT_UC8, 2400000.0, 0, &st);
    if (fn)
        p_cleanup_converter(st);
    (void)p_demodulate2400;
    (void)p_demodulate2400AC;
    printf("bind ok (converter %s)\n", fn ? "present" : "absent: no GPU");
    return 0;
}
