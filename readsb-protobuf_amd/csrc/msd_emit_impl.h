/* msd_emit_impl.h -- an accepted message as the msd_message record the host delivers, shared by the
 * stand-alone record kernel (msd_resolve_kernels.hip) and the scan kernel (msd_kernels.hip), whose
 * wavefronts write the previous batch's records on their way in.  Device code only. */
#ifndef MSD_EMIT_IMPL_H
#define MSD_EMIT_IMPL_H

#include "msd_internal.h"
#include "msd_kernels.h"

/* demod_2400.c:353-398 on the winning try of an accepted position.  signalLevel is the reference's own double
 * arithmetic (IEEE divisions, correctly rounded on this device as on the host: the build has no fast-math):
 * scaled_signal_power / 65535 / 65535 / signal_len.  `side` = the power sum with the length in the top 16 bits,
 * for the order-sensitive statistics the host keeps (demod_2400.c:399-408). */
__device__ __forceinline__ msd_message msd_emit_mode_s(const msd_acc rec, const msd_try *tries, unsigned long long power_sum,
                                                       uint64_t sample_ts, uint64_t sys_ts, uint32_t base,
                                                       unsigned long long &side)
{
    const msd_try *t = tries + rec.try_index;
    const uint4 lo = *reinterpret_cast<const uint4 *>(t);
    const uint4 hi = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(t) + 16);
    const uint32_t df = (lo.x & 0xffu) >> 3, tp = (lo.w >> 16) & 0xffu, errbit = lo.w >> 24;
    const uint32_t errbit2 = hi.w & 0xffu;
    const uint32_t msgbits = (df & 0x10u) ? 112u : 56u;
    const uint32_t j = rec.pos - base;
    msd_message mm;
    mm.timestampMsg = sample_ts + (uint64_t)j * 5 + (8 + 56) * 12 + tp;
    mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
    {
        const int signal_len = (int)(msgbits * 12u / 5u);
        const double signal_power = (double)power_sum / 65535.0 / 65535.0;
        mm.signalLevel = signal_power / signal_len;
        side = power_sum | ((unsigned long long)signal_len << 48);
    }
    mm.addr = hi.x; /* CRC for AP formats; AA after the fix otherwise (mode_s.c:559-562) */
    mm.crc = hi.y;
    mm.score = rec.score;
    mm.msgtype = (uint8_t)df;
    mm.msgbits = (uint8_t)msgbits;
    mm.correctedbits = errbit == 0xffu ? 0 : (errbit2 == 0xffu ? 1 : 2);
    mm.bestphase = (uint8_t)tp;
    uint32_t w[4] = {lo.x, lo.y, lo.z, lo.w & 0xffffu};
    if (errbit != 0xffu)
        w[errbit >> 5] ^= (0x80u >> (errbit & 7u)) << (8 * ((errbit >> 3) & 3u)); /* crc.c:417-425 */
    if (errbit2 != 0xffu)
        w[errbit2 >> 5] ^= (0x80u >> (errbit2 & 7u)) << (8 * ((errbit2 >> 3) & 3u));
#pragma unroll
    for (int k = 0; k < 14; ++k)
        mm.msg[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    mm.iid = df == 11 ? (uint8_t)(hi.y & 0x7fu) : 0;
    mm.pad = 0;
    return mm;
}

/* demod_2400.c:695-703, mode_ac.c:168-202 */
__device__ __forceinline__ msd_message msd_emit_mode_ac(const msd_ac_hit c, uint64_t sample_ts, uint64_t sys_ts)
{
    msd_message mm;
    mm.timestampMsg = sample_ts + c.f2_clock / 5; /* demod_2400.c:695 */
    mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
    mm.signalLevel = 0.0;
    mm.addr = (c.modeac & 0x0000FF7Fu) | (1u << 24);
    mm.crc = 0;
    mm.score = 0;
    mm.msgtype = 32;
    mm.msgbits = 16;
    mm.correctedbits = 0;
    mm.bestphase = 0;
#pragma unroll
    for (int k = 0; k < 14; ++k)
        mm.msg[k] = 0;
    mm.msg[0] = (uint8_t)(c.modeac >> 8);
    mm.msg[1] = (uint8_t)c.modeac;
    mm.iid = 0;
    mm.pad = 0;
    return mm;
}

#endif
