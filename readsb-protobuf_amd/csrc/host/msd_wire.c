/* msd_wire.c -- AVR raw lines and Beast binary frames of accepted messages (see msd_wire.h). */
#include "msd_wire.h"

#include <math.h>

static const char HEX[] = "0123456789ABCDEF";

size_t msd_avr_line(const msd_message *mm, int mlat, char *out)
{
    char *p = out;
    const int nbytes = mm->msgbits / 8;
    if (mlat && mm->timestampMsg) { /* net_io.c:877-881: the 48-bit counter, big-endian, 12 digits */
        *p++ = '@';
        for (int shift = 44; shift >= 0; shift -= 4)
            *p++ = HEX[(mm->timestampMsg >> shift) & 0xf];
    } else {
        *p++ = '*';
    }
    for (int j = 0; j < nbytes; ++j) {
        *p++ = HEX[mm->msg[j] >> 4];
        *p++ = HEX[mm->msg[j] & 0xf];
    }
    *p++ = ';';
    *p++ = '\n';
    *p = 0;
    return (size_t)(p - out);
}

size_t msd_beast_frame(const msd_message *mm, uint8_t *out)
{
    const int nbytes = mm->msgbits / 8;
    uint8_t body[6 + 1 + 14];
    uint8_t type;
    if (nbytes == 7)
        type = '2';
    else if (nbytes == 14)
        type = '3';
    else if (nbytes == 2)
        type = '1';
    else
        return 0; /* net_io.c:789-791 */
    for (int i = 0; i < 6; ++i)
        body[i] = (uint8_t)(mm->timestampMsg >> (40 - 8 * i));
    int sig = (int)round(sqrt(mm->signalLevel) * 255); /* net_io.c:819-823 */
    if (mm->signalLevel > 0 && sig < 1)
        sig = 1;
    if (sig > 255)
        sig = 255;
    body[6] = (uint8_t)sig;
    for (int j = 0; j < nbytes; ++j)
        body[7 + j] = mm->msg[j];
    uint8_t *p = out;
    *p++ = 0x1a;
    *p++ = type;
    for (int i = 0; i < 7 + nbytes; ++i) {
        *p++ = body[i];
        if (body[i] == 0x1a)
            *p++ = 0x1a;
    }
    return (size_t)(p - out);
}
