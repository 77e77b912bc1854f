/* msd_fifo.h -- the mag_buf FIFO of the host boundary; declared in include/modes_hip_readsb.h */
#ifndef MSD_FIFO_H
#define MSD_FIFO_H
#include "modes_hip_readsb.h"
#endif
