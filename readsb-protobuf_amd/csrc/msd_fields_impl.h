/* msd_fields_impl.h -- header fields of an accepted message, one implementation for the emit kernel
 * (device) and the host paths: what decodeModesMessage fills in after its CRC switch without looking
 * into the ME / MB payloads (mode_s.c:557-715), decodeAC13Field / decodeID13Field (mode_s.c:101-183),
 * the Gillham altitude of mode_ac.c:101-163, and decodeModeAMessage (mode_ac.c:168-202).
 * SURVEY.md 8(f) rank 1, first stage; extended squitter and Comm-B payloads are not decoded here. */
#ifndef MSD_FIELDS_IMPL_H
#define MSD_FIELDS_IMPL_H

#include <stdint.h>

#include "modes_hip.h"

#ifdef __HIPCC__
#define MSD_HD __host__ __device__ static inline
#else
#define MSD_HD static inline
#endif

/* bits first..last of the message, numbered from 1 at the most significant bit (mode_s.h:57-102);
 * at most 24 bits, which is all the header fields need */
MSD_HD uint32_t msd_field_bits(const uint8_t *msg, int first, int last)
{
    const int fb = (first - 1) >> 3, lb = (last - 1) >> 3;
    uint32_t v = 0;
    for (int i = fb; i <= lb; ++i)
        v = (v << 8) | msg[i];
    v >>= 7 - ((last - 1) & 7);
    return v & ((1u << (last - first + 1)) - 1u);
}

/* 13-bit identity code -> four octal digits, hex-coded A4A2A1 B4B2B1 C4C2C1 D4D2D1 (mode_s.c:101-143);
 * the X/M bit (0x0040) is dropped */
MSD_HD uint32_t msd_id13_to_squawk(uint32_t id13)
{
    uint32_t g = 0;
    g |= (id13 & 0x1000u) ? 0x0010u : 0; /* C1 */
    g |= (id13 & 0x0800u) ? 0x1000u : 0; /* A1 */
    g |= (id13 & 0x0400u) ? 0x0020u : 0; /* C2 */
    g |= (id13 & 0x0200u) ? 0x2000u : 0; /* A2 */
    g |= (id13 & 0x0100u) ? 0x0040u : 0; /* C4 */
    g |= (id13 & 0x0080u) ? 0x4000u : 0; /* A4 */
    g |= (id13 & 0x0020u) ? 0x0100u : 0; /* B1 */
    g |= (id13 & 0x0010u) ? 0x0001u : 0; /* D1 / Q */
    g |= (id13 & 0x0008u) ? 0x0200u : 0; /* B2 */
    g |= (id13 & 0x0004u) ? 0x0002u : 0; /* D2 */
    g |= (id13 & 0x0002u) ? 0x0400u : 0; /* B4 */
    g |= (id13 & 0x0001u) ? 0x0004u : 0; /* D4 */
    return g;
}

/* Gillham-coded Mode A value -> Mode C altitude in hundreds of feet, MSD_INVALID_ALTITUDE if it is
 * not one (mode_ac.c:78-163; the lookup there only keeps the twelve code bits of its argument) */
MSD_HD int32_t msd_mode_a_to_c(uint32_t mode_a)
{
    mode_a &= 0x7777u;
    if ((mode_a & 0x0001u) || !(mode_a & 0x00F0u)) /* D1 set is illegal, C1..C4 cannot all be zero */
        return MSD_INVALID_ALTITUDE;
    uint32_t hundreds = 0, five_hundreds = 0;
    if (mode_a & 0x0010u) hundreds ^= 7;   /* C1 */
    if (mode_a & 0x0020u) hundreds ^= 3;   /* C2 */
    if (mode_a & 0x0040u) hundreds ^= 1;   /* C4 */
    if ((hundreds & 5) == 5)
        hundreds ^= 2;                     /* 7 <-> 5 */
    if (hundreds > 5)
        return MSD_INVALID_ALTITUDE;
    if (mode_a & 0x0002u) five_hundreds ^= 0x0FF; /* D2 */
    if (mode_a & 0x0004u) five_hundreds ^= 0x07F; /* D4 */
    if (mode_a & 0x1000u) five_hundreds ^= 0x03F; /* A1 */
    if (mode_a & 0x2000u) five_hundreds ^= 0x01F; /* A2 */
    if (mode_a & 0x4000u) five_hundreds ^= 0x00F; /* A4 */
    if (mode_a & 0x0100u) five_hundreds ^= 0x007; /* B1 */
    if (mode_a & 0x0200u) five_hundreds ^= 0x003; /* B2 */
    if (mode_a & 0x0400u) five_hundreds ^= 0x001; /* B4 */
    if (five_hundreds & 1)
        hundreds = 6 - hundreds;
    return (int32_t)(five_hundreds * 5 + hundreds) - 13;
}

/* 13-bit altitude code (mode_s.c:152-183): feet; metric altitudes are not decoded */
MSD_HD int32_t msd_ac13_altitude(uint32_t ac13, uint8_t *unit)
{
    if (ac13 & 0x0040u) { /* M bit */
        *unit = 1;
        return MSD_INVALID_ALTITUDE;
    }
    *unit = 0;
    if (ac13 & 0x0010u) { /* Q bit: 25 ft steps */
        const int32_t n = (int32_t)(((ac13 & 0x1F80u) >> 2) | ((ac13 & 0x0020u) >> 1) | (ac13 & 0x000Fu));
        return n * 25 - 1000;
    }
    const int32_t n = msd_mode_a_to_c(msd_id13_to_squawk(ac13));
    if (n < -12)
        return MSD_INVALID_ALTITUDE;
    return 100 * n;
}

/* a Mode S message (msgtype 0..31, corrected bytes) */
MSD_HD void msd_fields_mode_s(const uint8_t *msg, uint32_t df, msd_fields *f)
{
    uint8_t *z = (uint8_t *)f;
    for (unsigned i = 0; i < sizeof *f; ++i)
        z[i] = 0;
    if (df == 0 || df == 4 || df == 16 || df == 20) { /* AC, mode_s.c:565-572 */
        f->AC = (uint16_t)msd_field_bits(msg, 20, 32);
        if (f->AC) {
            f->altitude_baro = msd_ac13_altitude(f->AC, &f->altitude_baro_unit);
            f->altitude_baro_valid = f->altitude_baro != MSD_INVALID_ALTITUDE;
        }
    }
    if (df == 11 || df == 17) { /* CA, mode_s.c:577-597 */
        f->CA = (uint8_t)msd_field_bits(msg, 6, 8);
        if (f->CA == 4)
            f->airground = 1;
        else if (f->CA == 5)
            f->airground = 2;
        else if (f->CA == 0 || f->CA == 6 || f->CA == 7)
            f->airground = 3;
    }
    if (df == 0)
        f->CC = (uint8_t)msd_field_bits(msg, 7, 7);
    if (df == 18)
        f->CF = (uint8_t)msd_field_bits(msg, 6, 8);
    if (df == 4 || df == 5 || df == 20 || df == 21) { /* DR, FS, UM: mode_s.c:610-650,703-705 */
        f->DR = (uint8_t)msd_field_bits(msg, 9, 13);
        f->UM = (uint8_t)msd_field_bits(msg, 14, 19);
        f->FS = (uint8_t)msd_field_bits(msg, 6, 8);
        f->alert_valid = f->spi_valid = f->FS <= 5;
        f->alert = f->FS == 2 || f->FS == 3 || f->FS == 4;
        f->spi = f->FS == 4 || f->FS == 5;
        if (f->FS <= 5)
            f->airground = (f->FS == 1 || f->FS == 3) ? 1 : 3;
    }
    if (df == 5 || df == 21) { /* ID, mode_s.c:653-660 */
        f->ID = (uint16_t)msd_field_bits(msg, 20, 32);
        if (f->ID) {
            f->squawk = (uint16_t)msd_id13_to_squawk(f->ID);
            f->squawk_valid = 1;
        }
    }
    if (df >= 24) { /* KE, ND: mode_s.c:663-665,692-694 */
        f->KE = (uint8_t)msd_field_bits(msg, 4, 4);
        f->ND = (uint8_t)msd_field_bits(msg, 5, 8);
    }
    if (df == 0 || df == 16) { /* RI, SL, VS: mode_s.c:697-716 */
        f->RI = (uint8_t)msd_field_bits(msg, 14, 17);
        f->SL = (uint8_t)msd_field_bits(msg, 9, 11);
        f->VS = (uint8_t)msd_field_bits(msg, 6, 6);
        f->airground = f->VS ? 1 : 3;
    }
}

/* a Mode A/C reply (mode_ac.c:168-202).  `carry` is the state demodulate2400AC's message record is in
 * when the reply is decoded: it is cleared once per buffer only (demod_2400.c:523-528), so a reply
 * that carries no altitude inherits altitude_baro / _valid / _unit from the last one of the same
 * buffer that did.  NULL = first reply of a buffer. */
MSD_HD void msd_fields_mode_ac(uint32_t mode_a, const msd_fields *carry, msd_fields *f)
{
    uint8_t *z = (uint8_t *)f;
    for (unsigned i = 0; i < sizeof *f; ++i)
        z[i] = 0;
    if (carry) {
        f->altitude_baro = carry->altitude_baro;
        f->altitude_baro_valid = carry->altitude_baro_valid;
        f->altitude_baro_unit = carry->altitude_baro_unit;
    }
    f->squawk = (uint16_t)(mode_a & 0x7777u);
    f->squawk_valid = 1;
    f->spi = (mode_a & 0x0080u) ? 1 : 0;
    f->spi_valid = 1;
    if (!f->spi) {
        const int32_t c = msd_mode_a_to_c(mode_a);
        if (c != MSD_INVALID_ALTITUDE) {
            f->altitude_baro = c * 100;
            f->altitude_baro_unit = 0;
            f->altitude_baro_valid = 1;
        }
    }
}

#endif
