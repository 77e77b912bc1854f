/*
 * msd_fifo.h -- the magnitude-buffer FIFO of the reference (fifo.h:57-120), same record and the
 * same eight calls, so a host written against fifo.h ports by renaming.  Two deliberate changes
 * (SURVEY.md 8(b), Appendix A.1):
 *   - fifo_enqueue's tail pointer is advanced, so a queue deeper than one buffer no longer drops
 *     buffers (fifo.c:192-197 never updates fifo_tail);
 *   - the timed waits test the return value of pthread_cond_timedwait against ETIMEDOUT
 *     (fifo.c:141,219 compare it with < 0, which never happens).
 */
#ifndef MSD_FIFO_H
#define MSD_FIFO_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MSD_MAGBUF_DISCONTINUOUS = 1, /* fifo.h:30-32 */
} msd_mag_buf_flags;

/* struct mag_buf, fifo.h:57-73 (field for field) */
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

bool msd_fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned overlap); /* fifo.h:80 */
void msd_fifo_destroy(void);                                                         /* fifo.h:84 */
void msd_fifo_drain(void);                                                           /* fifo.h:87 */
void msd_fifo_halt(void);                                                            /* fifo.h:94 */
struct msd_mag_buf *msd_fifo_acquire(uint32_t timeout_ms);                           /* fifo.h:99 */
void msd_fifo_enqueue(struct msd_mag_buf *buf);                                      /* fifo.h:111 */
struct msd_mag_buf *msd_fifo_dequeue(uint32_t timeout_ms);                           /* fifo.h:117 */
void msd_fifo_release(struct msd_mag_buf *buf);                                      /* fifo.h:120 */

#ifdef __cplusplus
}
#endif
#endif
