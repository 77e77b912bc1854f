/* msd_sdr_ifile.h -- the "ifile" SDR front-end of the host boundary; declared in include/modes_hip_readsb.h */
#ifndef MSD_SDR_IFILE_H
#define MSD_SDR_IFILE_H
#include "modes_hip_readsb.h"
#endif
