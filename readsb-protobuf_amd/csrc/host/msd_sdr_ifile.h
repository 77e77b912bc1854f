/*
 * msd_sdr_ifile.h -- the "ifile" SDR front-end of the reference (sdr_ifile.h / sdr.c:41-50), shaped
 * like its five-function handler so it can be registered in sdr_handlers[] (sdr.c:78-98):
 *     { msd_ifileInitConfig, msd_ifileHandleOption, msd_ifileOpen, msd_ifileRun, msd_ifileClose,
 *       "ifile", SDR_IFILE, 0 }
 * Two run modes:
 *   MSD_IFILE_MAGBUF  the literal drop-in: blocks of 131072 samples are converted on the GPU
 *                     (iq_convert_fn-shaped msd_convert), pushed through the mag_buf FIFO, and the
 *                     consumer calls the demodulate2400-shaped msd_demodulate_magbuf per buffer;
 *   MSD_IFILE_FUSED   the fast path: many blocks per call go straight to msd_submit_host, which runs
 *                     the fused convert+demodulate kernel; magnitudes never leave the GPU.
 * Both deliver the same ordered messages.
 */
#ifndef MSD_SDR_IFILE_H
#define MSD_SDR_IFILE_H

#include <stdbool.h>

#include "modes_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MSD_OPT_IFILE_NAME = 1, MSD_OPT_IFILE_FORMAT, MSD_OPT_IFILE_THROTTLE, MSD_OPT_IFILE_MODE }; /* help.h OptIfile* */
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
    int dc_filter;          /* Modes.dc_filter (--dcfilter, readsb.c:486); fused mode only */
} msd_receiver_options;

void msd_ifileInitConfig(void);                    /* sdr_ifile.c:70-80 */
bool msd_ifileHandleOption(int key, char *arg);    /* sdr_ifile.c:82-107 */
bool msd_ifileOpen(void);                          /* sdr_ifile.c:115-162 */
void msd_ifileRun(void);                           /* sdr_ifile.c:164-237 (blocks until EOF) */
void msd_ifileClose(void);                         /* sdr_ifile.c:239-255 */
void msd_ifileSetReceiver(const msd_receiver_options *opt);
int msd_ifileGetStats(msd_stats *st);
const char *msd_ifileLastError(void);

#ifdef __cplusplus
}
#endif
#endif
