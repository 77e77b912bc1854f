/* msd_wire.h -- the two wire formats readsb forwards accepted messages in:
 * AVR "raw" text lines (modesSendRawOutput, net_io.c:870-896; displayModesMessage --raw,
 * mode_s.c:1786-1798) and Beast binary frames (modesSendBeastOutput, net_io.c:769-835).
 * Plain C over msd_message, no state; the caller owns the buffers. */
#ifndef MSD_WIRE_H
#define MSD_WIRE_H

#include <stddef.h>
#include <stdint.h>

#include "modes_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_AVR_MAX 48   /* '@' + 12 hex + 28 hex + ';' + '\n' + NUL */
#define MSD_BEAST_MAX 44 /* 2 + 2 * (7 + 14): every byte after the type may need a 0x1A escape */

/* "*<hex>;\n", or "@<12 hex digits of the 12 MHz timestamp><hex>;\n" when mlat is set and the
 * timestamp is not zero (net_io.c:877-883).  Upper-case hex.  Returns the length (no NUL counted). */
size_t msd_avr_line(const msd_message *mm, int mlat, char *out);

/* 0x1A, type '1' (Mode A/C, 2 bytes) / '2' (56 bit) / '3' (112 bit), 6-byte big-endian timestamp,
 * signal byte round(sqrt(signalLevel) * 255) clamped to 1..255 for a non-zero level, payload; every
 * 0x1A after the type byte is doubled.  Returns the length, 0 for a message it cannot carry. */
size_t msd_beast_frame(const msd_message *mm, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
