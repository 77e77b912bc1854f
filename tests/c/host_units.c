/* The single-threaded host C under the sanitizers (scripts/sanitize.sh): wire formats, the constant tables with their
 * self-check and both CRC-repair tables, the field decoder on every DF with extreme payloads, the pacer, and the error
 * paths of the ifile handler and the converter factory on a box without a GPU. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "modes_hip_readsb.h"
#include "msd_internal.h"
#include "msd_wire.h"

int msd_tables_selftest(const msd_tables *t);

static uint32_t rs = 99;
static uint32_t rnd(void)
{
    rs ^= rs << 13;
    rs ^= rs >> 17;
    rs ^= rs << 5;
    return rs;
}

int main(void)
{
    /* tables */
    msd_tables *t = malloc(sizeof *t);
    for (int nfix = 0; nfix <= 2; ++nfix) {
        msd_tables_build(t, nfix);
        if (msd_tables_selftest(t))
            return 1;
    }
    for (int bits = 56; bits <= 112; bits += 56) {
        uint32_t lg = 0;
        uint64_t *f2 = msd_fix2_table(t, bits, &lg);
        if (!f2 || !lg)
            return 2;
        free(f2);
    }
    static const uint8_t frame[14] = {0x8D, 0x48, 0x40, 0xD6, 0x20, 0x2C, 0xC3, 0x71, 0xC3, 0x2C, 0xE0, 0x57, 0x60, 0x98};
    if (msd_crc24(t, frame, 112) != 0)
        return 3;
    uint16_t *q11 = malloc(sizeof(uint16_t) << 16);
    msd_sc16q11_table_build(8, q11);
    msd_sc16q11_table_build(1, q11);
    free(q11);
    free(t);

    /* wire formats and the field decoder: every DF, Mode A/C, zero / all-ones / random payloads and timestamps */
    char line[MSD_AVR_MAX];
    uint8_t beast[MSD_BEAST_MAX];
    size_t total = 0;
    msd_fields carry, out;
    memset(&carry, 0, sizeof carry);
    for (int k = 0; k < 20000; ++k) {
        msd_message mm;
        memset(&mm, 0, sizeof mm);
        const int kind = k % 3;
        for (int i = 0; i < 14; ++i)
            mm.msg[i] = kind == 0 ? 0 : kind == 1 ? 0xff : (uint8_t)rnd();
        mm.msgtype = (uint8_t)(k % 34); /* 0..31 DFs, 32 Mode A/C, 33: not a type at all */
        mm.msg[0] = (uint8_t)((mm.msgtype << 3) | (mm.msg[0] & 7));
        mm.msgbits = mm.msgtype == 32 ? 16 : (mm.msgtype & 16) ? 112 : 56;
        mm.timestampMsg = kind == 0 ? 0 : kind == 1 ? ~0ull : ((uint64_t)rnd() << 20) ^ rnd();
        mm.signalLevel = kind == 0 ? 0.0 : kind == 1 ? 1.0 : (rnd() & 0xffff) / 65536.0;
        mm.addr = rnd() & 0xffffff;
        mm.crc = rnd() & 0xffffff;
        if (k % 7 == 0)
            memset(mm.msg, 0x1a, sizeof mm.msg); /* the Beast escape byte everywhere */
        total += msd_avr_line(&mm, k & 1, line);
        total += msd_beast_frame(&mm, beast);
        msd_decode_fields(&mm, mm.msgtype == 32 ? &carry : NULL, &out);
        if (mm.msgtype == 32)
            carry = out;
    }
    if (!total)
        return 4;

    /* pacer: three buffers at 100 x real time */
    msd_pacer p;
    msd_pacer_start(&p, 2400000.0 * 100.0);
    for (int i = 0; i < 3; ++i)
        msd_pacer_wait(&p, 131072);

    /* the handler's and the factory's error paths (and, with a GPU, the happy ones) */
    msd_ifileInitConfig();
    if (msd_ifileOpen()) /* no file name */
        return 5;
    char name[] = "/nonexistent/capture.bin", fmt[] = "UC8";
    msd_ifileHandleOption(MSD_OPT_IFILE_NAME, name);
    msd_ifileHandleOption(MSD_OPT_IFILE_FORMAT, fmt);
    if (msd_ifileOpen())
        return 6;
    (void)msd_ifileLastError();
    msd_ifileClose();
    struct converter_state *st = NULL;
    msd_iq_convert_fn fn = msd_init_converter(MSD_INPUT_SC16, 2400000.0, 0, &st);
    if (fn) {
        static int16_t iq[2 * 4096];
        static uint16_t mag[4096];
        double lvl, pwr;
        for (int i = 0; i < 2 * 4096; ++i)
            iq[i] = (int16_t)rnd();
        fn(iq, mag, 4096, st, &lvl, &pwr);
        msd_cleanup_converter(st);
    }
    if (msd_init_converter((msd_input_format_t)7, 2400000.0, 0, &st))
        return 7;
    printf("host units ok (%zu wire bytes, converter %s)\n", total, fn ? "present" : "absent: no GPU");
    return 0;
}
