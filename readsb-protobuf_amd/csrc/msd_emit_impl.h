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

/* One wavefront writes its share of job J: wavefront w of the scan takes slice w % stride of buffer w / stride,
 * a run of consecutive records, between two of its tiles.  Up to 64 records at a time are put together in `lds` (3.5 KB of the wavefront's own) and leave as
 * consecutive dwords -- the destination is host memory, where a lane-strided struct store costs a PCIe write
 * per piece. */
__device__ inline void msd_emit_slice(const MsdEmitJob &J, uint32_t w, int lane, unsigned char *lds, bool dbg_no_store = false)
{
    const uint32_t b = w / J.stride, slice = w % J.stride;
    /* one round of loads: flags, the buffer's place in the record array (the power kernel's prefix), its counts, its clocks */
    const uint64_t ovf = J.totals[2], ac_ovf = J.ac ? J.ac_totals[2] : 0;
    const uint32_t o = J.rec_off[b], nm = J.nmsgs[b], na_all = J.ac ? J.nac[b] : 0u;
    if (ovf || ac_ovf)
        return; /* arenas overflowed: the host rescans the batch */
    const uint64_t sample_ts = J.ts[2 * b], sys_ts = J.ts[2 * b + 1];
    const uint32_t base = b * MSD_CHUNK_SAMPLES;
    const msd_acc *acc = J.acc + (size_t)b * MSD_RB_MSG_CAP;
    msd_wire *rec = reinterpret_cast<msd_wire *>(lds);
    static_assert(sizeof(msd_wire) % 8 == 0, "record size");
    auto flush = [&](uint32_t first, uint32_t n) { /* rows [first, first + n) from the LDS image, clipped to cap */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (first < J.cap && !dbg_no_store) {
            n = min(n, J.cap - first);
            unsigned long long *d = reinterpret_cast<unsigned long long *>(J.dense + first);
            const unsigned long long *r = reinterpret_cast<const unsigned long long *>(rec);
            for (uint32_t i = (uint32_t)lane; i < n * (uint32_t)(sizeof(msd_wire) / 8); i += 64)
                __builtin_nontemporal_store(r[i], &d[i]); /* streaming: nothing on the device reads it again */
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    {
        const uint32_t run = (nm + J.stride - 1) / J.stride, lo = min(nm, slice * run), hi = min(nm, lo + run);
        for (uint32_t m0 = lo; m0 < hi; m0 += 64) {
            const uint32_t m = m0 + (uint32_t)lane;
            if (m < hi) {
                unsigned long long side;
                rec[lane].mm = msd_emit_mode_s(acc[m], J.tries, J.power[(size_t)b * MSD_RB_MSG_CAP + m], sample_ts, sys_ts, base, side);
                if (o + m < J.cap)
                    J.side[o + m] = side;
            }
            flush(o + m0, min(64u, hi - m0));
        }
    }
    if (J.ac) { /* the buffer's Mode A/C replies follow its Mode S messages (readsb.c:826-829) */
        const uint32_t na = na_all;
        const uint32_t *acc_ac = J.acc_ac + (size_t)b * MSD_RB_AC_CAP;
        const uint32_t run = (na + J.stride - 1) / J.stride, lo = min(na, slice * run), hi = min(na, lo + run);
        for (uint32_t m0 = lo; m0 < hi; m0 += 64) {
            const uint32_t m = m0 + (uint32_t)lane;
            if (m < hi) {
                rec[lane].mm = msd_emit_mode_ac(J.ac[acc_ac[m]], sample_ts, sys_ts);
                if (o + nm + m < J.cap)
                    J.side[o + nm + m] = 0;
            }
            flush(o + nm + m0, min(64u, hi - m0));
        }
    }
}

#endif
