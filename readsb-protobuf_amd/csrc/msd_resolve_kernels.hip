/*
 * msd_resolve_kernels.hip -- the ordered resolve stage on the GPU: one workgroup per buffer replays
 * demodulate2400's state machine (skip-ahead, ICAO filter reads, accept/reject, counters,
 * demod_2400.c:236-428 + mode_s.c:311-409,424-555,717-726) over the candidate lists the scan
 * kernel left in HBM, against a snapshot of the ICAO filter.  The host keeps only what is really
 * sequential *across* buffers: replaying the ~50 filter adds per buffer and the 60 s flips to
 * decide which snapshot ("membership version") each buffer must see, and re-running the few
 * buffers that saw another one (msd_resolve.c explains why that converges to the exact result).
 *
 * Inside a buffer two things are sequential: the skip-ahead (an accepted message hides the next
 * 134/268 positions) and addresses the buffer itself adds.  Both are cheap once the expensive,
 * independent part -- hashing and probing the filter for every try, scoring, picking the best
 * phase -- has been done for all hits in parallel (phase P).  Phase S then walks the hits in order
 * on one lane, only re-evaluating a hit when one of the buffer's own *new* addresses (not in the
 * snapshot) could change its score; phase E builds the message records in parallel.
 */
#include <hip/hip_runtime.h>

#include "modes_hip.h"
#include "msd_internal.h"
#include "msd_kernels.h"
#include "msd_fields_impl.h"
#include "msd_emit_impl.h"
#include "msd_pred_impl.h"
#include "msd_mag_impl.h"

namespace {

#ifndef MSD_RESOLVE_WG
#define MSD_RESOLVE_WG 512
#endif
#ifndef MSD_POWER_PER
#define MSD_POWER_PER 5 /* messages whose samples a wavefront has in flight in the signal power step */
#endif
#ifndef MSD_RESOLVE_TIMING
#define MSD_RESOLVE_TIMING 0 /* 1: per-phase clocks of every workgroup in msd_rbuf.cyc (MSD_TRACE prints their means) */
#endif
#ifndef MSD_RESOLVE_OCC
#define MSD_RESOLVE_OCC 4 /* wavefronts per SIMD the resolve kernel is held to (two workgroups per CU; round 4: three per CU with
                             640- or 768-hit segments -- 50-56 KB of LDS, 80-90 registers -- were 4-5 % slower over the whole job) */
#endif
#ifndef MSD_RESOLVE_SEG
#define MSD_RESOLVE_SEG 1280
#endif
#ifndef MSD_RESOLVE_PRIO
#define MSD_RESOLVE_PRIO 0 /* s_setprio of the resolve wavefronts: matters only where they share a SIMD with scan wavefronts (LABLOG R5.7) */
#endif
constexpr int RT = MSD_RESOLVE_WG;   /* threads per workgroup (one workgroup per buffer) */
constexpr int SEG = MSD_RESOLVE_SEG; /* hits staged per segment */
constexpr uint32_t VACANT = 0xFFFFFFFFu;

/* A value every lane of the wavefront holds alike (read from one address): into scalar registers, so that it does not
 * occupy a vector register for as long as it lives and the branches on it are scalar. */
__device__ __forceinline__ uint32_t uni(uint32_t x)
{
    return __builtin_amdgcn_readfirstlane(x);
}
/* inclusive prefix sum over the wavefront (lane L: lanes 0..L; lane 63: the total): six DPP additions (row_shr 1, 2, 4,
 * 8, row_bcast 15 and 31) -- no LDS round trips, where a __shfl_up / __shfl_down ladder is six ds_bpermute, each waited
 * for */
__device__ __forceinline__ uint32_t wave_total_in_63(uint32_t v)
{
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ uint32_t wave_max_in_63(uint32_t v) /* the same ladder with max (a running maximum): missing lanes read 0 */
{
    uint32_t x = v;
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}
__device__ __forceinline__ uint64_t uni(uint64_t x)
{
    return (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)x) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32);
}
constexpr uint32_t SLOTS = 8192u;

__device__ __forceinline__ uint32_t hash24(uint32_t a) /* icao_filter.c:44-65 */
{
    uint32_t h = 0;
    h += a & 0xff;         h += h << 10; h ^= h >> 6;
    h += (a >> 8) & 0xff;  h += h << 10; h ^= h >> 6;
    h += (a >> 16) & 0xff; h += h << 10; h ^= h >> 6;
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return h & (SLOTS - 1);
}

/* linear probing from slot h; `stop` is the slot the probe started at (a full circle ends it).  The two tables of a
 * snapshot are interleaved (slot h of table k at word 2 h + k: one 8-byte load fetches both first slots); t points
 * at word k. */
__device__ __forceinline__ bool table_has(const uint32_t *t, uint32_t addr, uint32_t h, uint32_t stop)
{
    while (h != stop) {
        const uint32_t v = t[2 * h];
        if (v == addr)
            return true;
        if (v == VACANT)
            return false;
        h = (h + 1) & (SLOTS - 1);
    }
    return false;
}

/* icaoFilterTest (icao_filter.c:99-119) against a snapshot: two tables of 8192 slots.
 * bit 0: in table 0, bit 1: in table 1.  The first slot of both tables is fetched at once (at the
 * usual load that already decides both probes); only a collision walks on. */
__device__ __forceinline__ uint32_t snap_probe_from(const uint32_t *snap, uint32_t addr, uint32_t a, uint32_t b)
{
    const uint32_t start = hash24(addr); /* a, b: the first slot of the two tables, fetched by the caller */
    uint32_t where = 0;
    if (a == addr)
        where |= 1u;
    else if (a != VACANT && table_has(snap, addr, (start + 1) & (SLOTS - 1), start))
        where |= 1u;
    if (b == addr)
        where |= 2u;
    else if (b != VACANT && table_has(snap + 1, addr, (start + 1) & (SLOTS - 1), start))
        where |= 2u;
    return where;
}

struct TryView {
    uint32_t w0, w3; /* message bytes 0..3 and 12..15 (tp, errbit in the top half) */
    uint32_t addr, crc;
    uint32_t errbit2; /* 0xff: none */
};

__device__ __forceinline__ TryView load_try(const msd_try *t)
{
    /* the second half of the record: addr, crc, pos, { errbit2, msg[0], tp, errbit } */
    const uint4 hi = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(t) + 16);
    TryView v;
    v.w0 = (hi.w >> 8) & 0xffu;
    v.w3 = hi.w; /* tp and errbit sit where they do in the message's last word */
    v.addr = hi.x;
    v.crc = hi.y;
    v.errbit2 = hi.w & 0xffu;
    return v;
}

/* A try as phase P keeps it in LDS (everything the score and the verdict need):
 *   19 address known   20-24 DF   25-27 phase-4   28-29 corrected bits (0..2)   30 a corrected bit lies in AA
 *   36 DF11 with IID 0   37 address in the snapshot's active table   40-63 address
 * and the best phase of a hit: the same word plus  0-15 score (int16)   16-18 offset of the try. */
__device__ __forceinline__ uint64_t pack_try(const TryView &v, bool known, bool in_active)
{
    const uint32_t df = (v.w0 & 0xffu) >> 3, tp = (v.w3 >> 16) & 0xffu, errbit = v.w3 >> 24;
    const uint32_t nerr = errbit == 0xffu ? 0u : (v.errbit2 == 0xffu ? 1u : 2u);
    const bool in_aa = (errbit >= 8 && errbit <= 31) || (v.errbit2 >= 8 && v.errbit2 <= 31);
    return ((uint64_t)(known ? 1u : 0u) << 19) | ((uint64_t)df << 20) | ((uint64_t)(tp - 4) << 25) |
           ((uint64_t)(nerr | (in_aa ? 4u : 0u)) << 28) | ((uint64_t)((v.crc & 0x7fu) == 0 ? 1u : 0u) << 36) |
           ((uint64_t)(in_active ? 1u : 0u) << 37) | ((uint64_t)(v.addr & 0xffffffu) << 40);
}

/* scoreModesMessage (mode_s.c:311-409) on a packed try */
__device__ __forceinline__ int score_packed(uint64_t t)
{
    const uint32_t df = (uint32_t)(t >> 20) & 31u;
    const bool known = (t >> 19) & 1u;
    const int nerr = (int)((uint32_t)(t >> 28) & 3u);
    switch (df) {
    case 11: /* x / (nerr + 1) with nerr 0 or 1 (two wrong bits never get here): a shift */
        if ((t >> 36) & 1u)
            return (known ? 1600 : 750) >> nerr;
        return known ? 1000 >> nerr : -1;
    case 17: case 18:
        if (nerr == 2)
            return known ? 600 : 466; /* 1800 / 3, 1400 / 3 */
        return (known ? 1800 : 1400) >> nerr;
    case 20: case 21:
        return known ? 1000 : -2;
    default: /* 0, 4, 5, 16, 24: address/parity */
        return known ? 1000 : -1;
    }
}

__device__ __forceinline__ uint32_t res_len(uint64_t r) /* samples hidden by the message */
{
    return (((uint32_t)(r >> 20) & 0x10u) ? 112u : 56u) * 12u / 5u;
}

constexpr uint32_t TCAP = SEG + 3 * SEG / 10; /* tries staged per segment; a segment is cut short where they would not fit */
constexpr int MAXC = 5; /* a segment is made from at most MAXC x RT hits of the list */
constexpr uint32_t FCAP = 64;     /* new aircraft handled per round of a segment */
constexpr uint32_t ADDSET = 2048; /* > 2 x the 970 messages a buffer can hold */
#ifndef MSD_RESOLVE_SPEC
#define MSD_RESOLVE_SPEC 256 /* a power of two <= 256; the tests also run with 8 (table-full paths) */
#endif
constexpr uint32_t SPEC = MSD_RESOLVE_SPEC; /* new aircraft of one buffer followed speculatively (the rest go the slow way) */
static_assert(SPEC >= 2 && SPEC <= 256 && (SPEC & (SPEC - 1)) == 0, "hash is eight bits");

__device__ __forceinline__ bool addset_has(const uint32_t *addset, uint32_t addr)
{
    uint32_t hs = (addr * 2654435761u) >> 21;
    for (;;) {
        const uint32_t v = addset[hs];
        if (v == addr)
            return true;
        if (v == VACANT)
            return false;
        hs = (hs + 1) & (ADDSET - 1);
    }
}

__device__ __forceinline__ int spec_find(const uint32_t *key, uint32_t addr)
{
    uint32_t h = ((addr * 2654435761u) >> 24) & (SPEC - 1);
    for (uint32_t probes = 0; probes < SPEC; ++probes) {
        const uint32_t k = key[h];
        if (k == addr)
            return (int)h;
        if (k == VACANT)
            return -1;
        h = (h + 1) & (SPEC - 1);
    }
    return -1;
}

/* Signal power of the accepted messages of one buffer, by the workgroup that accepted them: a wavefront per message,
 * MSD_POWER_PER messages' samples in flight per wavefront (the step is a chain of dependent loads -- IQ bytes,
 * magnitude table -- and nothing else).  The wavefront's records (message w, w + 8, w + 16, ...) are fetched once, two
 * per lane, and handed round by v_readlane.  Written in stages -- every address, every IQ load, every table load, then the sums -- so that the loads
 * of a round are in flight together whatever the register allocator makes of the rest of the kernel; a round whose
 * messages lie wholly inside the batch (all but those in its first 300 samples) addresses the IQ array directly. */
template <int FMT>
__device__ inline void power_of_accepted(const MsdResolveParams &P, const msd_acc *acc, uint32_t nm, unsigned long long *out,
                                         int tid)
{
    MsdSampleSource S;
    S.iq = P.iq;
    S.prev_tail = P.prev_tail;
    S.have_prev = P.have_prev;
    S.batch_first = P.batch_first;
    S.nsamples = P.nsamples;
    constexpr uint32_t PER = MSD_POWER_PER, NW = RT / 64;
    constexpr uint32_t BPS = (FMT == MSD_FMT_SC16 || FMT == MSD_FMT_SC16Q11) ? 4 : 2;
    constexpr uint32_t QN = (MSD_RB_MSG_CAP + 64 * NW - 1) / (64 * NW); /* records per lane: two for eight wavefronts, four for four */
    static_assert(QN == 2 || QN == 4, "two or four records per lane");
    const int lane = tid & 63;
    const uint32_t wave = (uint32_t)(tid >> 6);
    if (wave >= nm)
        return;
    uint32_t rpos[QN], rlen[QN];
#pragma unroll
    for (uint32_t q = 0; q < QN; ++q) {
        const uint32_t m = wave + ((uint32_t)lane + 64u * q) * NW;
        const msd_acc rec = acc[m < nm ? m : wave];
        rpos[q] = rec.pos;
        rlen[q] = m < nm ? rec.len : 0u;
    }
    for (uint32_t q0 = 0; wave + q0 * NW < nm; q0 += PER) { /* wave-uniform */
        const uint32_t m0 = wave + q0 * NW;
        uint32_t x[PER][5], pos[PER], len[PER];
        bool inside = true;
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            const uint32_t q = q0 + u; /* < 64 QN + PER: past the last record the lengths are zero */
            uint32_t pu = 0, lu = 0;
#pragma unroll
            for (uint32_t k = 0; k < QN; ++k) {
                const uint32_t pk = __builtin_amdgcn_readlane(rpos[k], q & 63u), lk = __builtin_amdgcn_readlane(rlen[k], q & 63u);
                pu = (q >> 6) == k ? pk : pu;
                lu = (q >> 6) == k ? lk : lu;
            }
            pos[u] = q < 64u * QN ? pu : __builtin_amdgcn_readlane(rpos[QN - 1], q & 63u);
            len[u] = lu;
            const int64_t rel0 = (int64_t)pos[u] - (int64_t)MSD_OVERLAP + 19;
            inside = inside && rel0 >= 0 && rel0 + (int64_t)len[u] <= (int64_t)S.nsamples;
        }
        if (inside) {
#pragma unroll
            for (uint32_t u = 0; u < PER; ++u) {
                const uint32_t rel0 = pos[u] - MSD_OVERLAP + 19u;
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const uint32_t k = (uint32_t)lane + 64u * v;
                    const uint32_t off = (rel0 + (k < len[u] ? k : 0u)) * BPS;
                    if (BPS == 2)
                        x[u][v] = *reinterpret_cast<const uint16_t *>(S.iq + off);
                    else
                        x[u][v] = *reinterpret_cast<const uint32_t *>(S.iq + off);
                }
            }
            if (FMT == MSD_FMT_UC8) {
#pragma unroll
                for (uint32_t u = 0; u < PER; ++u)
#pragma unroll
                    for (int v = 0; v < 5; ++v)
                        x[u][v] = P.lut[fold8(x[u][v] >> 8) * MSD_LUT_STRIDE + fold8(x[u][v] & 0xffu)];
            } else if (FMT != MSD_FMT_MAG16) {
                const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
#pragma unroll
                for (uint32_t u = 0; u < PER; ++u)
#pragma unroll
                    for (int v = 0; v < 5; ++v)
                        x[u][v] = mag_from_s16((int)(int16_t)(x[u][v] & 0xffffu), (int)(int16_t)(x[u][v] >> 16), inv);
            }
#pragma unroll
            for (uint32_t u = 0; u < PER; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v)
                    x[u][v] = ((uint32_t)lane + 64u * v) < len[u] ? x[u][v] : 0u;
        } else {
#pragma unroll
            for (uint32_t u = 0; u < PER; ++u)
                msd_power_loads<FMT>(S, P.lut, pos[u], len[u], lane, x[u]);
        }
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            /* at most 320 squares < 2^32 each: two 32-bit sums (low and high halves of the squares) */
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const uint32_t sq = x[u][v] * x[u][v];
                lo += sq & 0xffffu;
                hi += sq >> 16;
            }
            lo = wave_total_in_63(lo);
            hi = wave_total_in_63(hi);
            const uint32_t m = m0 + u * NW;
            if (lane == 63 && m < nm)
                out[m] = (unsigned long long)lo + ((unsigned long long)hi << 16);
        }
    }
}

__global__ void __launch_bounds__(RT, MSD_RESOLVE_OCC) msd_resolve_kernel(const MsdResolveParams P)
{
    __shared__ msd_hit seg_hits[SEG];
    __shared__ uint64_t seg_res[SEG];
    __shared__ uint64_t seg_try[TCAP];
    __shared__ uint16_t seg_toff[SEG + 1];
    __shared__ uint16_t seg_thit[TCAP];       /* owner of each staged try */
    __shared__ uint32_t out_short[MSD_RB_ADD_INLINE]; /* those the host must apply: not in the active table yet */
    __shared__ uint32_t addset[ADDSET]; /* addresses this buffer has passed to icaoFilterAdd */
    __shared__ uint32_t okb[SEG / 32];  /* hits that would be accepted if nothing hides them */
    __shared__ uint16_t ok_idx[SEG], ok_next[SEG], acc_k[SEG];
    uint16_t *const add_first = ok_next; /* per accepted message of a round; the chain walk is over by then */
    __shared__ uint32_t add_rank[ADDSET]; /* per address-set slot: rank of the first message of the round that adds it */
    __shared__ uint32_t ok_pos[SEG];
    __shared__ uint32_t front[SEG];     /* where the scan resumes once it is past hit i (relative to the buffer) */
    __shared__ uint32_t sh_ctr[16];
    /* New aircraft inside the buffer: address -> position (<< 1 | long message) of the buffer's first CRC-clean DF17 /
     * DF11 (IID 0) candidate with an address no filter knows.  Tries of that address behind that message are staged as
     * known right away -- the message will add the address if it is accepted, and a clean one nearly always is --
     * instead of being found out round by round; every round checks the assumption against the messages it accepted
     * (spec_conf: 1 the message was accepted and adds the address, 2 it was not: the tries go back to unknown). */
    __shared__ uint32_t spec_key[SPEC], spec_val[SPEC];
    __shared__ uint8_t spec_conf[SPEC];
    __shared__ uint32_t sh_nspec, sh_nuse, sh_specfail;
    __shared__ uint32_t sh_wsum[RT / 64];
    __shared__ __attribute__((aligned(16))) uint32_t sh_cw[MAXC][RT / 64]; /* per chunk and wavefront: hits with tries | tries << 16 */
    __shared__ uint64_t sh_range[2];
    __shared__ __attribute__((aligned(16))) uint32_t sh_rpre[64]; /* lean layout: hits of the buffer in front of each of its regions */
    __shared__ uint64_t sh_resume, sh_seg_resume, sh_now;
    __shared__ uint32_t sh_nmsgs, sh_nadds, sh_nshort, sh_next, sh_nfit, sh_nok, sh_na, sh_last, sh_nf;
    __shared__ uint32_t f_addr[FCAP], f_resume[FCAP], f_idx[FCAP]; /* new aircraft of the round: address, end and hit of the adding message */

    const int tid = threadIdx.x;
    const uint32_t wave = uni((uint32_t)tid >> 6);
    if (MSD_RESOLVE_PRIO)
        __builtin_amdgcn_s_setprio(MSD_RESOLVE_PRIO);
    uint32_t cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = wall_clock64();
#if MSD_RESOLVE_TIMING
#define PHASE(k) { const uint64_t n_ = wall_clock64(); cyc[k] += (uint32_t)(n_ - tlast); tlast = n_; }
#else
#define PHASE(k) { (void)tlast; }
#endif
    /* everything the set-up needs from global memory, in three rounds of independent loads */
    const uint64_t ovf = uni((uint64_t)P.totals[2]), nhits = uni((uint64_t)P.totals[0]);
    const uint64_t ac_ovf = P.ac ? uni((uint64_t)P.ac_totals[2]) : 0;
    /* First pass: every buffer in turn against snapshot 0, the clocks a linear function of the buffer index
     * (sdr_ifile.c:187-190) -- worked out here instead of read from the control arrays, which live in host memory
     * (two dependent reads over PCIe before the workgroup knows which buffer it has). */
    const bool implicit = P.first_pass && P.ctl_implicit;
    const uint32_t b = implicit ? blockIdx.x : uni(P.todo[blockIdx.x]);
    const uint32_t npred_raw = uni(msd_pred_count(P.pred, P.pred_gen));
    if (P.first_pass && P.region_counts) {
        /* lean layout: what the gather kernel used to leave for the host -- the buffer's level / power sums (the
         * device cells are zeroed for the slot's next batch) and, from the first workgroup, the batch's totals */
        if (tid < 2) {
            P.h_sums[2 * b + tid] = P.sums[2 * b + tid];
            P.sums[2 * b + tid] = 0;
            if (P.fmeans)
                P.h_fmeans[2 * b + tid] = P.fmeans[2 * b + tid];
        }
        if (blockIdx.x == 0 && tid >= 64 && tid < 68 && P.ac)
            P.h_ac_totals[tid - 64] = P.ac_totals[tid - 64];
        if (blockIdx.x == 0 && tid < 64) {
            unsigned long long h = 0, t = 0;
            for (uint32_t i = tid; i < P.nscan_wg; i += 64) {
                h += P.wg_totals[i].nhits;
                t += P.wg_totals[i].ntries;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                h += __shfl_down(h, d, 64);
                t += __shfl_down(t, d, 64);
            }
            if (tid == 0) {
                P.h_totals[0] = h;
                P.h_totals[1] = t;
                P.h_totals[2] = ovf;
            }
        }
    }
    if (ovf || ac_ovf)
        return; /* the candidate arenas overflowed: the host rescans the batch in pieces */
    if (blockIdx.x == 0 && P.first_pass) { /* the prediction list of the batch, for the host's replay */
        const uint32_t n = npred_raw;
        const uint32_t *list = reinterpret_cast<const uint32_t *>(P.pred + MSD_PRED_SLOTS) + 2;
        if (tid == 0)
            *P.h_pred_count = n <= MSD_PRED_LIST ? n : MSD_PRED_LIST + 1;
        for (uint32_t i = tid; i < n && i < MSD_PRED_LIST; i += RT) {
            const uint32_t h = list[i];
            const unsigned long long en = P.pred[h];
            msd_pred_entry e;
            e.addr = (uint32_t)(en >> 32) & 0xffffffu;
            e.first = (uint32_t)en;
            e.slot = h;
            e.pad = 0;
            P.h_pred[i] = e;
        }
    }
    const uint32_t snap_index = implicit ? 0u : uni(P.snap_idx[b]);
    const uint32_t *snap = P.snaps + (size_t)snap_index * MSD_SNAP_WORDS;
    uint32_t mlen;
    uint64_t sample_ts_b, sys_ts;
    if (implicit) {
        const uint64_t first = (uint64_t)b * MSD_CHUNK_SAMPLES;
        const uint64_t left = P.batch_samples > first ? P.batch_samples - first : 0;
        mlen = (uint32_t)(left > MSD_CHUNK_SAMPLES ? MSD_CHUNK_SAMPLES : left);
        /* the host's own expression, in doubles like sdr_ifile.c:187 (beyond 7.5e8 samples the product is no longer
         * exact, and the truncation must fall the same way) */
        sample_ts_b = (uint64_t)((double)(P.sample_counter0 + first) * 12e6 / 2400000.0);
        sys_ts = sample_ts_b / 12000u;
    } else {
        mlen = uni(P.valid[b]);
        sample_ts_b = uni((uint64_t)P.ts[2 * b]);
        sys_ts = uni((uint64_t)P.ts[2 * b + 1]);
    }
    const uint32_t snap_active = uni(snap[2 * SLOTS]) & 1u;
    const uint64_t base = (uint64_t)b * MSD_CHUNK_SAMPLES, end = base + mlen;
    msd_acc *acc = P.acc + (size_t)b * MSD_RB_MSG_CAP;
    uint32_t *adds = P.adds + (size_t)b * MSD_RB_MSG_CAP;
    msd_rbuf *rb = P.rbuf + b;
    for (int i = tid; i < (int)ADDSET; i += RT)
        addset[i] = VACANT;
    if (tid < 16)
        sh_ctr[tid] = 0;
    if (tid < (int)SPEC) {
        spec_key[tid] = VACANT;
        spec_val[tid] = ~0u;
        spec_conf[tid] = 0;
    }
    if (tid == 0)
        sh_nspec = 0;
    if (P.region_counts) { /* lean layout: the buffer's hits are the slices of its k regions, one after the other */
        if (tid < 64) {
            const uint32_t k = P.regions_per_buffer;
            uint32_t c = 0;
            if ((uint32_t)tid < k && b * k + (uint32_t)tid < P.nregions)
                c = min(P.region_counts[(size_t)b * k + tid].nhits, P.hcap);
            const uint32_t incl = wave_total_in_63(c);
            sh_rpre[tid] = incl - c; /* hits in the buffer's earlier regions */
            if (tid == 63) {
                sh_range[0] = 0;
                sh_range[1] = incl;
            }
        }
    } else if (P.buf_first) { /* the gather kernel left the range of this buffer's hits in the ordered list */
        if (tid < 2)
            sh_range[tid] = min((uint64_t)P.buf_first[b + tid], nhits);
    } else if (tid < 128) { /* ... or a 64-ary search per wavefront */
        const int lane = tid & 63;
        const uint64_t want = base + ((tid >> 6) ? MSD_CHUNK_SAMPLES : 0);
        uint64_t lo = 0, hi = nhits; /* lower_bound lies in [lo, hi] */
        while (lo < hi) {
            const uint64_t step = (hi - lo + 63) / 64;
            const uint64_t p = lo + (uint64_t)lane * step;
            const bool below = p < hi && MSD_HIT_POS(P.hits[p]) < want;
            const int k = __popcll(__ballot(below)); /* the first k probes are below: monotone */
            if (k == 0) {
                hi = lo;
            } else {
                const uint64_t nhi = lo + (uint64_t)k * step;
                lo = lo + (uint64_t)(k - 1) * step + 1;
                if (nhi < hi)
                    hi = nhi;
            }
        }
        if (lane == 0)
            sh_range[tid >> 6] = lo;
    }
    if (tid == 0) {
        sh_resume = base;
        sh_now = sys_ts; /* demod_2400.c:252-255 */
        sh_nmsgs = sh_nadds = sh_nshort = 0;
    }
    __syncthreads();
    const uint64_t hb = uni(sh_range[0]), he = uni(sh_range[1]);
    PHASE(0)

    /* hit v of the buffer's ordered list */
    uint32_t rp[8]; /* lean layout, up to eight regions per buffer: their places in the list, in scalar registers */
    {
        const uint4 r0 = *reinterpret_cast<const uint4 *>(&sh_rpre[0]), r1 = *reinterpret_cast<const uint4 *>(&sh_rpre[4]);
        const uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w}; /* two LDS reads, then the lot */
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q)
            rp[q] = (P.region_counts && q < P.regions_per_buffer) ? uni(r[q]) : ~0u;
    }
    auto hit_at = [&](uint64_t v) -> msd_hit {
        if (P.region_counts) {
            const uint32_t k = P.regions_per_buffer;
            uint32_t j = 0, first = 0; /* the last region that starts at or before v, and where it starts */
            if (k <= 8) {
#pragma unroll
                for (uint32_t q = 1; q < 8; ++q) {
                    const bool le = rp[q] <= (uint32_t)v; /* ascending: true for a prefix of the regions */
                    j = le ? q : j;
                    first = le ? rp[q] : first;
                }
            } else {
                for (uint32_t q = 1; q < k; ++q)
                    j = sh_rpre[q] <= (uint32_t)v ? q : j;
                first = sh_rpre[j];
            }
            return P.hits[((size_t)b * k + j) * P.hcap + ((uint32_t)v - first)];
        }
        return P.hits[v];
    };
    /* A segment holds the hits that have tries, in order.  A hit without one (no phase of the preamble got past the
     * DF / CRC tests) scores -2 whatever the filter holds: it only counts, unless a message hides it -- the thread that
     * loaded it keeps it in a register with the number of segment hits in front of it, and looks at the resume
     * frontier front[] once the segment is done. */
    uint32_t nraw = 0;
    for (uint64_t s0 = hb; s0 < he; s0 += nraw) {
        msd_hit keep[MAXC];
        uint32_t keep_d[MAXC], place[MAXC]; /* keep_d: position (17 bits) | phases (3) | segment hits in front (12) */
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { /* up to MAXC x RT hits of the list, all loads first: one round trip */
            const uint64_t v = s0 + (uint64_t)c * RT + tid;
            keep[c] = hit_at(v < he ? v : he - 1);
        }
        /* place of every hit among those with tries, and of its tries among the segment's: inside the wavefront here,
         * the wavefronts' sums through LDS */
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const uint64_t v = s0 + (uint64_t)c * RT + tid;
            const uint32_t nl = v < he ? MSD_HIT_NLIVE(keep[c]) : 0u;
            const uint64_t bal = __ballot(nl != 0);
            const uint32_t lrank = (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
            const uint32_t incl = wave_total_in_63(nl);
            place[c] = lrank | ((incl - nl) << 8); /* rank among the wavefront's hits with tries | its earlier lanes' tries */
            if ((tid & 63) == 63)
                sh_cw[c][tid >> 6] = (uint32_t)__popcll(bal) | (incl << 16);
        }
        if (tid == 0) {
            sh_seg_resume = sh_resume;
            sh_nfit = ~0u;
        }
        __syncthreads();
        /* the sums of the wavefronts in front of mine, chunk by chunk: every wavefront runs the same 40-entry scan in
         * its lanes (eight lanes a chunk; both counts in one word, neither reaches 2^16) and picks its two numbers per
         * chunk out with v_readlane */
        constexpr int NWV = RT / 64; /* wavefronts per workgroup: 8, or 4 for the slim workgroup that fits beside a scan workgroup */
        static_assert((NWV == 8 || NWV == 4) && MAXC * NWV <= 64, "one lane per chunk and wavefront");
        uint32_t cw_incl, cw_excl;
        {
            const uint32_t x = (tid & 63) < MAXC * NWV ? (&sh_cw[0][0])[tid & 63] : 0u;
            uint32_t incl = x;
            {
                const uint32_t u1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false); /* row_shr:1 */
                incl += (tid & (NWV - 1)) >= 1 ? u1 : 0u;
                const uint32_t u2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
                incl += (tid & (NWV - 1)) >= 2 ? u2 : 0u;
                if (NWV == 8) {
                    const uint32_t u4 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
                    incl += (tid & 7) >= 4 ? u4 : 0u;
                }
            }
            cw_incl = incl;
            cw_excl = incl - x;
        }
        uint32_t n = 0, ntr = 0;
        nraw = 0;
        bool open = true; /* uniform */
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            keep_d[c] = ~0u;
            const uint32_t tot = __builtin_amdgcn_readlane(cw_incl, NWV * c + NWV - 1), mine = __builtin_amdgcn_readlane(cw_excl, NWV * c + wave);
            const uint32_t ch = tot & 0xffffu, ct = tot >> 16; /* hits with tries / tries of the chunk ... */
            const uint32_t bh = mine & 0xffffu, bt = mine >> 16; /* ... and of the wavefronts in front of mine */
            const uint64_t v = s0 + (uint64_t)c * RT + tid;
            open = open && s0 + (uint64_t)c * RT < he && n + ch <= (uint32_t)SEG; /* the first chunk always fits: RT <= SEG */
            if (open) {
                const msd_hit h = keep[c];
                const uint32_t nl = v < he ? MSD_HIT_NLIVE(h) : 0u;
                const uint32_t d = n + bh + (place[c] & 0xffu), toff = ntr + bt + (place[c] >> 8);
                if (nl) {
                    seg_hits[d] = h;
                    ok_idx[d] = (uint16_t)(c * RT + tid); /* place in the list, should the segment be cut short below */
                    seg_toff[d] = (uint16_t)toff;
                    if (toff + nl > TCAP) /* cut the segment in front of the first hit whose tries do not fit */
                        atomicMin(&sh_nfit, d);
                    else
                        for (uint32_t q = 0; q < nl; ++q)
                            seg_thit[toff + q] = (uint16_t)d;
                } else if (v < he && MSD_HIT_POS(h) < end) {
                    keep_d[c] = (uint32_t)(MSD_HIT_POS(h) - base) | (MSD_HIT_MASK(h) << 17) | (d << 20);
                }
                n += ch;
                ntr += ct;
                nraw += (he - (s0 + (uint64_t)c * RT) < (uint64_t)RT) ? (uint32_t)(he - (s0 + (uint64_t)c * RT)) : (uint32_t)RT;
            }
        }
        if (tid == 0)
            seg_toff[n] = (uint16_t)ntr; /* entry n = all tries of the segment (nobody's hit: d < n) */
        __syncthreads();
        if (uni(sh_nfit) < n) {
            n = uni(sh_nfit); /* >= 1: a hit's tries always fit; entry n of seg_toff is that hit's own */
            nraw = uni((uint32_t)ok_idx[n]); /* the next segment starts with the hit that did not fit */
        }
        const uint32_t ntries = uni((uint32_t)seg_toff[n]);
        PHASE(5)
        /* ---- phase P: stage every try of the segment in LDS with its filter verdict: up to NT tries per thread, in
         * stages -- the tries, then the first slots of the tables (the prediction table's beside the snapshot's two),
         * then the verdicts -- so that a stage's loads are one round trip ---- */
        {
            constexpr uint32_t NT = 3; /* per round */
            for (uint32_t t0 = 0; t0 < ntries; t0 += NT * RT) {
                TryView tv[NT];
                uint32_t sa[NT], sb[NT];
                unsigned long long pe[NT];
#pragma unroll
                for (uint32_t q = 0; q < NT; ++q) {
                    const uint32_t t = t0 + (uint32_t)tid + q * RT;
                    const uint32_t tt = t < ntries ? t : t0;
                    const uint32_t i = seg_thit[tt];
                    tv[q] = load_try(P.tries + MSD_HIT_TRY(seg_hits[i]) + (tt - seg_toff[i]));
                }
#pragma unroll
                for (uint32_t q = 0; q < NT; ++q) {
                    const uint32_t start = hash24(tv[q].addr);
                    const uint2 ab = *reinterpret_cast<const uint2 *>(snap + 2 * start);
                    sa[q] = ab.x;
                    sb[q] = ab.y;
                    pe[q] = npred_raw ? P.pred[MSD_PRED_HASH(tv[q].addr) & (MSD_PRED_SLOTS - 1)] : ~0ull;
                }
#pragma unroll
                for (uint32_t q = 0; q < NT; ++q) {
                    const uint32_t t = t0 + (uint32_t)tid + q * RT;
                    if (t < ntries) {
                        const TryView &v = tv[q];
                        const uint32_t where = snap_probe_from(snap, v.addr, sa[q], sb[q]);
                        bool known = where != 0 || addset_has(addset, v.addr);
                        if (!known && npred_raw) { /* added by an earlier buffer of this batch (predicted; the host verifies) */
                            const unsigned long long key = msd_pred_key(P.pred_gen, v.addr);
                            if ((pe[q] & 0xffffffff00000000ull) == key)
                                known = (uint32_t)pe[q] < b;
                            else if ((uint32_t)(pe[q] >> 56) == P.pred_gen)
                                known = msd_pred_lookup(P.pred, P.pred_gen, v.addr) < b;
                        }
                        seg_try[t] = pack_try(v, known, (where >> snap_active) & 1u);
                        const uint32_t df = (v.w0 & 0xffu) >> 3;
                        if (!known && (v.w3 >> 24) == 0xffu && (df == 17 || (df == 11 && (v.crc & 0x7fu) == 0))) {
                            /* this message would add the address (mode_s.c:717-726): note the first one per address */
                            const uint32_t at = (uint32_t)(MSD_HIT_POS(seg_hits[seg_thit[t]]) - base);
                            uint32_t h = ((v.addr * 2654435761u) >> 24) & (SPEC - 1);
                            for (uint32_t probes = 0; probes < SPEC; ++probes) {
                                const uint32_t old = atomicCAS(&spec_key[h], VACANT, v.addr);
                                if (old == VACANT || old == v.addr) {
                                    atomicMin(&spec_val[h], (at << 1) | (df == 17 ? 1u : 0u));
                                    if (old == VACANT)
                                        atomicAdd(&sh_nspec, 1u);
                                    break;
                                }
                                h = (h + 1) & (SPEC - 1);
                            }
                        }
                    }
                }
            }
        }
        if (tid == 0)
            sh_nuse = 0;
        __syncthreads();
        if (uni(sh_nspec)) { /* tries of those addresses behind the message that will add them: known, on probation */
            for (uint32_t t = tid; t < ntries; t += RT) {
                const uint64_t v = seg_try[t];
                if ((v >> 19) & 1u)
                    continue;
                const int e = spec_find(spec_key, (uint32_t)(v >> 40));
                if (e < 0)
                    continue;
                const uint32_t val = spec_val[e];
                const uint32_t resume = (val >> 1) + ((val & 1u) ? 268u : 134u) + 1u;
                if ((uint32_t)(MSD_HIT_POS(seg_hits[seg_thit[t]]) - base) >= resume) {
                    seg_try[t] = v | (1ull << 19) | (1ull << 39);
                    sh_nuse = 1; /* benign race: everybody writes 1 */
                }
            }
        }
        __syncthreads();
        PHASE(1)

        uint32_t start = 0;
        while (start < n) {
            if (tid < (int)(SEG / 32))
                okb[tid] = 0;
            if (tid == 0)
                sh_last = 0;
            __syncthreads();
            /* best phase of every remaining hit: strict '>' so the first-tried phase wins ties
             * (demod_2400.c:218); a position without tries scores -2.  Then the acceptance part of
             * decodeModesMessage (mode_s.c:424-555), which only needs what the score used.
             * seg_res bit 38: would be accepted if no earlier message hides it. */
            for (uint32_t i = start + tid; i < n; i += RT) {
                const msd_hit h = seg_hits[i];
                const uint32_t nlive = MSD_HIT_NLIVE(h), o = seg_toff[i];
                front[i] = 0;
                int bestscore = -2;
                uint64_t best = 0;
                for (uint32_t q = 0; q < nlive; ++q) {
                    const uint64_t t = seg_try[o + q];
                    const int sc = score_packed(t);
                    if (sc > bestscore) {
                        bestscore = sc;
                        best = t | ((uint64_t)q << 16);
                    }
                }
                uint64_t r = best | ((uint64_t)bestscore & 0xffffu);
                if (bestscore >= 0 && MSD_HIT_POS(h) < end) {
                    const uint32_t known = (uint32_t)(r >> 19) & 1u, df = (uint32_t)(r >> 20) & 31u;
                    const bool nerr = ((uint32_t)(r >> 28) & 3u) != 0, in_aa = (r >> 30) & 1u;
                    bool reject;
                    if (df == 11)
                        reject = nerr && !known; /* mode_s.c:492-498 */
                    else if (df == 17 || df == 18)
                        reject = in_aa && !known; /* mode_s.c:522-526: the fix changed AA */
                    else
                        reject = !known;
                    if (!reject) {
                        r |= 1ull << 38;
                        atomicOr(&okb[i >> 5], 1u << (i & 31));
                    }
                }
                seg_res[i] = r;
            }
            __syncthreads();
            PHASE(2)
            /* ---- phase S: which of the acceptable hits are not hidden by an earlier accepted one ----
             * ordered list of the acceptable hits ... */
            if (tid < 64) {
                const uint32_t w = tid < (int)(SEG / 32) ? tid : 0;
                uint32_t m = tid < (int)(SEG / 32) ? okb[w] : 0u;
                if (w == (start >> 5))
                    m &= ~0u << (start & 31);
                else if (w < (start >> 5))
                    m = 0;
                uint32_t incl = (uint32_t)__popc(m);
                const uint32_t cnt = incl;
                incl = wave_total_in_63(cnt);
                uint32_t k = incl - cnt;
                while (m) {
                    const uint32_t i = w * 32 + (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    ok_idx[k] = (uint16_t)i;
                    ok_pos[k] = (uint32_t)(MSD_HIT_POS(seg_hits[i]) - base);
                    ++k;
                }
                if (tid == 63)
                    sh_nok = incl;
            }
            __syncthreads();
            const uint32_t nok = uni(sh_nok);
            /* ... for each, the first acceptable hit behind its message (bit 15: it adds an address the
             * filter does not know yet, so the hits behind it have to be looked at again) ... */
            for (uint32_t k = tid; k < nok; k += RT) {
                const uint64_t r = seg_res[ok_idx[k]];
                const uint32_t resume = ok_pos[k] + res_len(r) + 1; /* j += len, then the loop's ++ */
                uint32_t lo = k + 1, hi = nok;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ok_pos[mid] < resume)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                const uint32_t df = (uint32_t)(r >> 20) & 31u, nerr = (uint32_t)(r >> 28) & 3u;
                const bool adds_addr = nerr == 0 && (df == 17 || (df == 11 && ((r >> 36) & 1u))); /* mode_s.c:717-726 */
                const bool fresh = adds_addr && !((r >> 19) & 1u);
                ok_next[k] = (uint16_t)(lo | (fresh ? 0x8000u : 0u));
            }
            __syncthreads();
            /* ... and the chain of accepted ones, the only sequential bit.  A message that adds an
             * address the filter does not know yet (a new aircraft) may change how later hits score;
             * the chain runs on and the check below finds the first hit that really is affected. */
            if (tid < 64) {
                /* One wavefront: a window of 64 chain pointers per LDS read, walked by v_readlane (a few cycles a step
                 * instead of an LDS round trip); the lanes of the messages passed write them out together. */
                const uint32_t from = uni((uint32_t)(sh_resume - base));
                uint32_t lo = 0, hi = nok;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (uni(ok_pos[mid]) < from)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                uint32_t k = lo, na = 0, next = n, nf = 0;
                bool full = false;
                while (k < nok && !full) {
                    const uint32_t w0 = k;
                    const uint32_t mine = w0 + (uint32_t)tid < nok ? (uint32_t)ok_next[w0 + tid] : 0x7fffu;
                    const uint64_t fresh = __ballot((mine & 0x8000u) != 0); /* messages that add a new aircraft */
                    const uint32_t nxt = mine & 0x7fffu;
                    const uint32_t lim = nok - w0 < 64u ? nok - w0 : 64u; /* >= 1 */
                    uint64_t passed = 0;
                    uint32_t off = 0; /* the walk proper: one v_readlane and a handful of scalar instructions a message */
                    do {
                        passed |= 1ull << off;
                        off = __builtin_amdgcn_readlane(nxt, off) - w0;
                    } while (off < lim); /* past the window or the list */
                    k = w0 + off;
                    for (uint64_t fp = passed & fresh; fp; fp &= fp - 1) { /* in order */
                        const uint32_t bit = (uint32_t)__builtin_ctzll(fp), kk = w0 + bit;
                        const uint32_t i = uni((uint32_t)ok_idx[kk]);
                        { /* the message the probation table is waiting for: every later try of its address was staged as
                           * known already (and the ones under the message are hidden), nothing is left to look for */
                            const int e = spec_find(spec_key, uni((uint32_t)(seg_res[i] >> 40)));
                            if (e >= 0 && uni(spec_val[e] >> 1) == uni(ok_pos[kk]) && uni((uint32_t)spec_conf[e]) != 2u)
                                continue; /* (an entry that failed its probation has had its tries put back to unknown: the message
                                             is a new aircraft like any other, should it be accepted after all) */
                        }
                        if (nf == FCAP) { /* more new aircraft than the check handles at once: stop in front of this one */
                            next = i;
                            passed &= (1ull << bit) - 1ull;
                            full = true;
                            break;
                        }
                        if (tid == 0) {
                            const uint64_t r = seg_res[i];
                            f_addr[nf] = (uint32_t)(r >> 40);
                            f_resume[nf] = ok_pos[kk] + res_len(r) + 1;
                            f_idx[nf] = i;
                        }
                        ++nf;
                    }
                    if ((passed >> tid) & 1ull)
                        acc_k[na + (uint32_t)__popcll(passed & ((1ull << tid) - 1ull))] = (uint16_t)(w0 + tid);
                    na += (uint32_t)__popcll(passed);
                }
                if (tid == 0) {
                    sh_na = na;
                    sh_next = next;
                    sh_nf = nf;
                }
            }
            __syncthreads();
            const uint32_t nf = uni(sh_nf), nuse = uni(sh_nuse);
            if (nf | nuse) { /* uniform */
                if (nf) {
                    /* first hit with a not yet known try of one of the new addresses behind the message that adds it */
                    uint32_t first = uni(sh_next);
                    for (uint32_t t = tid; t < ntries; t += RT) {
                        const uint64_t v = seg_try[t];
                        if ((v >> 19) & 1u)
                            continue;
                        const uint32_t i = seg_thit[t];
                        if (i < start || i >= first)
                            continue;
                        const uint32_t addr = (uint32_t)(v >> 40), pos = (uint32_t)(MSD_HIT_POS(seg_hits[i]) - base);
                        for (uint32_t f = 0; f < nf; ++f)
                            if (f_addr[f] == addr && pos >= f_resume[f]) {
                                first = i;
                                break;
                            }
                    }
                    if (first < sh_next)
                        atomicMin(&sh_next, first);
                }
                if (tid == 0)
                    sh_specfail = 0;
                __syncthreads();
                if (nuse) {
                    /* the tries on probation: the messages of this round (in front of the cut so far) that add their
                     * address confirm the table's entries ... */
                    const uint32_t cut1 = uni(sh_next), na1 = uni(sh_na);
                    for (uint32_t j = tid; j < na1; j += RT) {
                        const uint32_t k = acc_k[j], i = ok_idx[k];
                        if (i >= cut1)
                            continue;
                        const uint64_t r = seg_res[i];
                        const uint32_t df = (uint32_t)(r >> 20) & 31u, nerr = (uint32_t)(r >> 28) & 3u;
                        if (nerr == 0 && (df == 17 || (df == 11 && ((r >> 36) & 1u)))) {
                            const int e = spec_find(spec_key, (uint32_t)(r >> 40));
                            if (e >= 0 && (spec_val[e] >> 1) == ok_pos[k])
                                spec_conf[e] = 1;
                        }
                    }
                    __syncthreads();
                    /* ... a try in front of the cut whose entry is not confirmed (the message was looked at and is not
                     * among the accepted ones) was staged wrongly: the round ends in front of the first such hit ... */
                    uint32_t first = cut1;
                    for (uint32_t t = tid; t < ntries; t += RT) {
                        const uint64_t v = seg_try[t];
                        if (!((v >> 39) & 1u))
                            continue;
                        const uint32_t i = seg_thit[t];
                        if (i < start || i >= cut1)
                            continue;
                        const int e = spec_find(spec_key, (uint32_t)(v >> 40));
                        if (e >= 0 && spec_conf[e] == 1)
                            continue;
                        if (e >= 0)
                            spec_conf[e] = 2;
                        first = i < first ? i : first;
                        sh_specfail = 1; /* benign race */
                    }
                    if (first < cut1)
                        atomicMin(&sh_next, first);
                    __syncthreads();
                    if (uni(sh_specfail)) { /* ... and all tries of such an address go back to unknown */
                        for (uint32_t t = tid; t < ntries; t += RT) {
                            const uint64_t v = seg_try[t];
                            if (!((v >> 39) & 1u))
                                continue;
                            const int e = spec_find(spec_key, (uint32_t)(v >> 40));
                            if (e < 0 || spec_conf[e] == 2)
                                seg_try[t] = v & ~((1ull << 19) | (1ull << 39));
                        }
                        __syncthreads();
                    }
                }
                const uint32_t cut = uni(sh_next);
                if (nuse) {
                    /* A confirmation by a message that the cut has dropped since is void (round 6: the fuzzer's case 702780 -- the
                     * probation check above moved the cut in front of a message that had confirmed its entry a few lines earlier;
                     * re-evaluated, another message hid it, it was never accepted, and every later try of its address stayed
                     * "known" on the strength of a round that did not happen).  The message is looked at again behind the cut. */
                    const uint32_t na1 = uni(sh_na);
                    for (uint32_t j = tid; j < na1; j += RT) {
                        const uint32_t k = acc_k[j], i = ok_idx[k];
                        if (i < cut)
                            continue;
                        const uint64_t r = seg_res[i];
                        const uint32_t df = (uint32_t)(r >> 20) & 31u, nerr = (uint32_t)(r >> 28) & 3u;
                        if (nerr == 0 && (df == 17 || (df == 11 && ((r >> 36) & 1u)))) {
                            const int e = spec_find(spec_key, (uint32_t)(r >> 40));
                            if (e >= 0 && (spec_val[e] >> 1) == ok_pos[k] && spec_conf[e] == 1)
                                spec_conf[e] = 0;
                        }
                    }
                    __syncthreads();
                }
                /* the new addresses whose messages stay accepted are known from here on */
                if (nf) {
                    for (uint32_t t = tid; t < ntries; t += RT) {
                        const uint64_t v = seg_try[t];
                        if ((v >> 19) & 1u)
                            continue;
                        const uint32_t addr = (uint32_t)(v >> 40);
                        for (uint32_t f = 0; f < nf; ++f)
                            if (f_addr[f] == addr && f_idx[f] < cut) {
                                seg_try[t] = v | (1ull << 19);
                                break;
                            }
                    }
                }
                if (tid == 0) { /* drop the accepted messages at or behind the cut */
                    uint32_t na = sh_na;
                    while (na && ok_idx[acc_k[na - 1]] >= cut)
                        --na;
                    sh_na = na;
                }
                __syncthreads();
            }
            if (tid == 0 && sh_na) {
                const uint32_t lastk = acc_k[sh_na - 1];
                sh_resume = base + ok_pos[lastk] + res_len(seg_res[ok_idx[lastk]]) + 1;
            }
            __syncthreads();
            PHASE(3)
            /* the accepted messages of this round, in parallel: records, counters, addresses to add */
            const uint32_t na = uni(sh_na), nmsgs0 = uni(sh_nmsgs);
            for (uint32_t j = tid; j < na; j += RT) {
                const uint32_t i = ok_idx[acc_k[j]];
                const msd_hit h = seg_hits[i];
                const uint64_t r = seg_res[i];
                const uint32_t df = (uint32_t)(r >> 20) & 31u, nerr = (uint32_t)(r >> 28) & 3u;
                const uint32_t addr = (uint32_t)(r >> 40);
                front[i] = ok_pos[acc_k[j]] + res_len(r) + 1u; /* j += len, then the loop's ++ */
                if (nmsgs0 + j < MSD_RB_MSG_CAP) {
                    msd_acc rec;
                    rec.pos = (uint32_t)MSD_HIT_POS(h);
                    rec.try_index = (uint32_t)(MSD_HIT_TRY(h) + ((uint32_t)(r >> 16) & 7u));
                    rec.score = (int32_t)(int16_t)(r & 0xffffu);
                    rec.len = res_len(r);
                    acc[nmsgs0 + j] = rec;
                }
                atomicAdd(&sh_ctr[3 + nerr], 1u);
                atomicAdd(&sh_ctr[11 + ((uint32_t)(r >> 25) & 7u)], 1u);
                add_first[j] = 0;
                if (nerr == 0 && (df == 17 || (df == 11 && ((r >> 36) & 1u)))) { /* mode_s.c:717-726 */
                    uint32_t hs = (addr * 2654435761u) >> 21;
                    for (;;) {
                        const uint32_t old = atomicCAS(&addset[hs], VACANT, addr);
                        if (old == VACANT) { /* new in this buffer; several lanes may hold the same address: */
                            add_rank[hs] = 0xffffffffu;
                            break;
                        }
                        if (old == addr)
                            break;
                        hs = (hs + 1) & (ADDSET - 1);
                    }
                    add_first[j] = (uint16_t)(hs | 0x8000u); /* slot of the address; resolved below */
                }
            }
            __syncthreads();
            /* the first message of the round with a given new address is the one that adds it */
            for (uint32_t j = tid; j < na; j += RT) {
                const uint32_t f = add_first[j];
                if (f & 0x8000u) {
                    const uint32_t hs = f & 0x7fffu;
                    if (add_rank[hs] != 0xfffffffeu) /* added in an earlier round or segment */
                        atomicMin(&add_rank[hs], j);
                }
            }
            __syncthreads();
            { /* append them in message order (the host applies them in this order): thread t looks at
                 * messages PERJ t .. PERJ t + PERJ - 1, a workgroup-wide exclusive count gives the places
                 * (adds in the low half, those the host filter does not know yet in the high half) */
                constexpr uint32_t PERJ = (SEG + RT - 1) / RT;
                const uint32_t nadds0 = uni(sh_nadds), nshort0 = uni(sh_nshort);
                uint32_t a_addr[PERJ], mine = 0;
                bool a_is[PERJ], a_short[PERJ];
#pragma unroll
                for (uint32_t q = 0; q < PERJ; ++q) {
                    const uint32_t j = (uint32_t)tid * PERJ + q;
                    a_is[q] = a_short[q] = false;
                    a_addr[q] = 0;
                    if (j < na) {
                        const uint32_t f = add_first[j];
                        if ((f & 0x8000u) && add_rank[f & 0x7fffu] == j) {
                            const uint64_t r = seg_res[ok_idx[acc_k[j]]];
                            a_is[q] = true;
                            a_short[q] = !((r >> 37) & 1u); /* not in the active table yet: the host filter changes */
                            a_addr[q] = (uint32_t)(r >> 40);
                            add_rank[f & 0x7fffu] = 0xfffffffeu; /* nobody else's j equals the old value */
                        }
                    }
                    mine += (a_is[q] ? 1u : 0u) + (a_short[q] ? 0x10000u : 0u);
                }
                const uint32_t incl = wave_total_in_63(mine);
                if ((tid & 63) == 63)
                    sh_wsum[tid >> 6] = incl;
                __syncthreads();
                uint32_t off = incl - mine;
                for (int w = 0; w < (tid >> 6); ++w)
                    off += sh_wsum[w];
                if (tid == RT - 1) { /* off + mine = the round's totals */
                    const uint32_t tot = off + mine;
                    sh_nadds = nadds0 + (tot & 0xffffu);
                    sh_nshort = nshort0 + (tot >> 16);
                    sh_nmsgs = nmsgs0 + na;
                }
#pragma unroll
                for (uint32_t q = 0; q < PERJ; ++q) {
                    if (a_is[q]) {
                        const uint32_t ia = nadds0 + (off & 0xffffu);
                        if (ia < MSD_RB_MSG_CAP)
                            adds[ia] = a_addr[q]; /* the complete list stays in device memory */
                        if (a_short[q]) {
                            const uint32_t is = nshort0 + (off >> 16);
                            if (is < MSD_RB_ADD_INLINE)
                                out_short[is] = a_addr[q];
                        }
                    }
                    off += (a_is[q] ? 1u : 0u) + (a_short[q] ? 0x10000u : 0u);
                }
            }
            __syncthreads();
            { /* the resume frontier: an inclusive running maximum over the segment (the entries of this round's messages
               * were set above, those of earlier rounds are final, the rest is zero) */
                constexpr uint32_t PERF = (SEG + RT - 1) / RT;
                uint32_t f[PERF], run = 0;
#pragma unroll
                for (uint32_t k = 0; k < PERF; ++k) {
                    const uint32_t i = (uint32_t)tid * PERF + k;
                    const uint32_t x = i < n ? front[i] : 0u;
                    run = x > run ? x : run;
                    f[k] = run;
                }
                const uint32_t incl = wave_max_in_63(run);
                if ((tid & 63) == 63)
                    sh_wsum[tid >> 6] = incl;
                uint32_t pre = __shfl_up(incl, 1, 64);
                if ((tid & 63) == 0)
                    pre = 0;
                __syncthreads();
                for (int w = 0; w < (tid >> 6); ++w) {
                    const uint32_t x = sh_wsum[w];
                    pre = x > pre ? x : pre;
                }
#pragma unroll
                for (uint32_t k = 0; k < PERF; ++k) {
                    const uint32_t i = (uint32_t)tid * PERF + k;
                    if (i < n)
                        front[i] = f[k] > pre ? f[k] : pre;
                }
                __syncthreads();
            }
            /* ---- phase C: the counters of every hit that no accepted message hides, in parallel ---- */
            const uint32_t stop_at = uni(sh_next);
            const uint32_t segres = uni((uint32_t)(sh_seg_resume - base));
            uint32_t c_pre = 0, c_bad = 0, c_unk = 0, c_p01 = 0, c_p23 = 0, c_p4 = 0, last = 0;
            for (uint32_t i = start + tid; i < stop_at; i += RT) {
                const msd_hit h = seg_hits[i];
                const uint64_t a = MSD_HIT_POS(h);
                if (a >= end)
                    continue;
                uint32_t resume = i ? front[i - 1] : 0u;
                resume = resume > segres ? resume : segres;
                if ((uint32_t)(a - base) < resume)
                    continue;
                const uint32_t mask = MSD_HIT_MASK(h);
                c_pre++;
                c_p01 += mask & 1u;
                c_p23 += (mask >> 1) & 1u;
                c_p4 += (mask >> 2) & 1u;
                const uint64_t r = seg_res[i];
                const uint32_t sc = (uint32_t)r & 0xffffu;
                c_bad += sc == 0xfffeu;
                /* unknown: scored -1, or scored >= 0 and then failed the acceptance rules */
                c_unk += (sc == 0xffffu) || (!(sc & 0x8000u) && !((r >> 38) & 1u));
                if (!(sc & 0x8000u))
                    last = i + 1; /* reached decodeModesMessage: it set Modes.ifile_now first */
            }
            { /* six counters of at most 4 per lane, packed ten bits apart, summed over the wavefront */
                const uint32_t pk_lo = wave_total_in_63(c_pre | (c_bad << 10) | (c_unk << 20));
                const uint32_t pk_hi = wave_total_in_63(c_p01 | (c_p23 << 10) | (c_p4 << 20));
                const uint64_t pk = (uint64_t)pk_lo | ((uint64_t)pk_hi << 30);
                last = wave_max_in_63(last);
                if ((tid & 63) == 63 && pk) {
                    atomicAdd(&sh_ctr[0], (uint32_t)pk & 1023u);
                    atomicAdd(&sh_ctr[1], (uint32_t)(pk >> 10) & 1023u);
                    atomicAdd(&sh_ctr[2], (uint32_t)(pk >> 20) & 1023u);
                    const uint32_t p01 = (uint32_t)(pk >> 30) & 1023u, p23 = (uint32_t)(pk >> 40) & 1023u;
                    atomicAdd(&sh_ctr[6], p01);
                    atomicAdd(&sh_ctr[7], p01);
                    atomicAdd(&sh_ctr[8], p23);
                    atomicAdd(&sh_ctr[9], p23);
                    atomicAdd(&sh_ctr[10], (uint32_t)(pk >> 50) & 1023u);
                    atomicMax(&sh_last, last);
                }
            }
            start = stop_at;
            __syncthreads();
            if (tid == 0 && sh_last) { /* demod_2400.c:358-366: (timestampMsg - sampleTimestamp) / 12000 ms */
                const uint32_t i = sh_last - 1;
                const uint64_t r = seg_res[i];
                const uint32_t tp = 4 + ((uint32_t)(r >> 25) & 7u);
                sh_now = sys_ts + ((uint32_t)(MSD_HIT_POS(seg_hits[i]) - base) * 5u + (8 + 56) * 12 + tp) / 12000u;
            }
            PHASE(4)
        }
        { /* the segment's hits without a try: counted unless a message hides them (the scan resumes behind a message,
           * demod_2400.c:404-410); front[] is complete for the segment now */
            const uint32_t segres = uni((uint32_t)(sh_seg_resume - base));
            uint32_t d_pre = 0, d_p01 = 0, d_p23 = 0, d_p4 = 0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const uint32_t kd = keep_d[c], d = kd >> 20;
                if (kd == ~0u || (uint32_t)(c * RT + tid) >= nraw || d > n)
                    continue;
                uint32_t resume = d ? front[d - 1] : 0u;
                resume = resume > segres ? resume : segres;
                if ((kd & 0x1ffffu) < resume)
                    continue;
                const uint32_t mask = (kd >> 17) & 7u;
                d_pre++;
                d_p01 += mask & 1u;
                d_p23 += (mask >> 1) & 1u;
                d_p4 += (mask >> 2) & 1u;
            }
            /* every hit without a try that counted is a "bad" rejection (score -2) */
            d_pre = wave_total_in_63(d_pre);
            d_p01 = wave_total_in_63(d_p01);
            d_p23 = wave_total_in_63(d_p23);
            d_p4 = wave_total_in_63(d_p4);
            if ((tid & 63) == 63 && d_pre) {
                atomicAdd(&sh_ctr[0], d_pre);
                atomicAdd(&sh_ctr[1], d_pre);
                atomicAdd(&sh_ctr[6], d_p01);
                atomicAdd(&sh_ctr[7], d_p01);
                atomicAdd(&sh_ctr[8], d_p23);
                atomicAdd(&sh_ctr[9], d_p23);
                atomicAdd(&sh_ctr[10], d_p4);
            }
            __syncthreads(); /* front[] and the segment arrays are rewritten by the next segment */
        }
        PHASE(7)
    }

    if (P.power) { /* the signal power of the buffer's messages (the records in acc[] are this workgroup's own) */
        __syncthreads();
        const uint32_t nm = uni(sh_nmsgs) < MSD_RB_MSG_CAP ? uni(sh_nmsgs) : MSD_RB_MSG_CAP;
        unsigned long long *out = P.power + (size_t)b * MSD_RB_MSG_CAP;
        switch (P.format) {
        case MSD_FMT_UC8: power_of_accepted<MSD_FMT_UC8>(P, acc, nm, out, tid); break;
        case MSD_FMT_SC16: power_of_accepted<MSD_FMT_SC16>(P, acc, nm, out, tid); break;
        case MSD_FMT_SC16Q11: power_of_accepted<MSD_FMT_SC16Q11>(P, acc, nm, out, tid); break;
        default: power_of_accepted<MSD_FMT_MAG16>(P, acc, nm, out, tid); break;
        }
        PHASE(6)
    }

    /* ---- Mode A/C (demod_2400.c:522-708): every test is done, only the 69-sample skip-ahead of an
     * accepted reply is left (:705); no filter, independent of the Mode S messages ---- */
    uint32_t nac_total = 0;
    if (P.ac) {
        const uint64_t nac_all = P.ac_totals[0];
        if (tid < 128) { /* the buffer's range in the ordered candidate list, as above */
            const int lane = tid & 63;
            const uint64_t want = base + ((tid >> 6) ? MSD_CHUNK_SAMPLES : 0);
            uint64_t lo = 0, hi = nac_all;
            while (lo < hi) {
                const uint64_t step = (hi - lo + 63) / 64;
                const uint64_t p = lo + (uint64_t)lane * step;
                const bool below = p < hi && P.ac[p].pos < want;
                const int k = __popcll(__ballot(below));
                if (k == 0) {
                    hi = lo;
                } else {
                    const uint64_t nhi = lo + (uint64_t)k * step;
                    lo = lo + (uint64_t)(k - 1) * step + 1;
                    if (nhi < hi)
                        hi = nhi;
                }
            }
            if (lane == 0)
                sh_range[tid >> 6] = lo;
        }
        if (tid == 0)
            sh_next = 0; /* first sample not hidden by an accepted reply, relative to the buffer */
        __syncthreads();
        const uint64_t ab = sh_range[0], ae = sh_range[1];
        uint32_t *acc_ac = P.acc_ac + (size_t)b * MSD_RB_AC_CAP;
        uint32_t m = 0;
        for (uint64_t s0 = ab; s0 < ae; s0 += m) {
            m = (ae - s0 < (uint64_t)SEG) ? (uint32_t)(ae - s0) : (uint32_t)SEG;
            for (uint32_t i = tid; i < m; i += RT)
                ok_pos[i] = (uint32_t)(P.ac[s0 + i].pos - base);
            __syncthreads();
            for (uint32_t k = tid; k < m; k += RT) { /* first candidate behind reply k: f1_sample += 20*87/25, then ++ */
                const uint32_t resume = ok_pos[k] + 69u + 1u;
                uint32_t lo = k + 1, hi = m;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ok_pos[mid] < resume)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                ok_next[k] = (uint16_t)lo;
            }
            __syncthreads();
            if (tid == 0) {
                const uint32_t from = sh_next;
                uint32_t lo = 0, hi = m;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ok_pos[mid] < from)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                uint32_t k = lo, na = 0, resume = from;
                while (k < m && ok_pos[k] < mlen) {
                    acc_k[na++] = (uint16_t)k;
                    resume = ok_pos[k] + 70u;
                    k = ok_next[k];
                }
                sh_na = na;
                sh_next = resume;
            }
            __syncthreads();
            const uint32_t na = sh_na;
            for (uint32_t j = tid; j < na; j += RT)
                if (nac_total + j < MSD_RB_AC_CAP)
                    acc_ac[nac_total + j] = (uint32_t)(s0 + acc_k[j]);
            nac_total += na;
            __syncthreads();
        }
    }

    { /* flush the lists, coalesced */
        const uint32_t ns = sh_nshort < MSD_RB_ADD_INLINE ? sh_nshort : MSD_RB_ADD_INLINE;
        for (uint32_t i = tid; i < ns; i += RT)
            rb->adds[i] = out_short[i];
    }
    if (tid == 0) {
        for (int k = 0; k < 16; ++k)
            rb->ctr[k] = sh_ctr[k];
        rb->nmsgs = sh_nmsgs;
        rb->nadds = sh_nadds;
        rb->nshort = sh_nshort;
        rb->nac = nac_total;
        rb->version_used = snap_index;
        rb->fallback = (sh_nmsgs > MSD_RB_MSG_CAP || sh_nadds > MSD_RB_MSG_CAP || nac_total > MSD_RB_AC_CAP) ? 1u : 0u;
        rb->end_now = sh_now;
        for (int k = 0; k < 8; ++k)
            rb->cyc[k] = cyc[k];
        P.nmsgs[b] = sh_nmsgs < MSD_RB_MSG_CAP ? sh_nmsgs : MSD_RB_MSG_CAP;
        if (P.ac)
            P.nac[b] = nac_total < MSD_RB_AC_CAP ? nac_total : MSD_RB_AC_CAP;
    }
}

/* rows [first, first + n) of the record arrays from their LDS images, clipped to cap */
__device__ inline void emit_rows(msd_wire *dense, msd_fields *fields, const msd_wire *sh_rec, const msd_fields *sh_f,
                                 uint32_t first, uint32_t n, uint32_t cap)
{
    if (first >= cap)
        return;
    n = min(n, cap - first);
    static_assert(sizeof(msd_wire) % 8 == 0 && sizeof(msd_fields) % 4 == 0, "record sizes");
    uint2 *d = reinterpret_cast<uint2 *>(dense + first);
    const uint2 *r = reinterpret_cast<const uint2 *>(sh_rec);
    for (uint32_t i = threadIdx.x; i < n * (uint32_t)(sizeof(msd_wire) / 8); i += blockDim.x)
        d[i] = r[i];
    if (fields) {
        uint32_t *fd = reinterpret_cast<uint32_t *>(fields + first);
        const uint32_t *fr = reinterpret_cast<const uint32_t *>(sh_f);
        for (uint32_t i = threadIdx.x; i < n * (uint32_t)(sizeof(msd_fields) / 4); i += blockDim.x)
            fd[i] = fr[i];
    }
}

/* The accepted messages of buffer b as 56-byte msd_message records, dense over the batch, written straight to
 * page-locked host memory (msd_emit_impl.h builds them; `side` gets the power sum and length per record, for the
 * host's statistics).  This kernel is the record writer with MSD_CFG_DECODE_FIELDS (FIELDS = true: the header
 * fields, 140 bytes more per message), in the side-stream layout, for the last batch of a run and after a
 * second resolve pass; in the in-order layout the wavefronts of the next scan write the records
 * (msd_emit_slice, MsdScanParams.emit). */
template <bool FIELDS>
__global__ void __launch_bounds__(256) msd_emit_kernel(const MsdResolveParams P, const unsigned long long *power,
                                                       unsigned long long *side, msd_wire *dense, msd_fields *fields_arg,
                                                       uint32_t cap)
{
    msd_fields *const fields = FIELDS ? fields_arg : nullptr;
    if (P.totals[2] || (P.ac && P.ac_totals[2]))
        return;
    const uint32_t b = blockIdx.x;
    /* o = messages (Mode S and Mode A/C) in front of this buffer */
    __shared__ uint32_t part[4];
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < b; i += 256)
        mine += P.nmsgs[i] + (P.ac ? P.nac[i] : 0u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = mine;
    __syncthreads();
    const uint32_t o = part[0] + part[1] + part[2] + part[3], nm = P.nmsgs[b];
    const uint64_t sample_ts = P.ts[2 * b], sys_ts = P.ts[2 * b + 1];
    const uint32_t base = b * MSD_CHUNK_SAMPLES;
    const msd_acc *acc = P.acc + (size_t)b * MSD_RB_MSG_CAP;
    /* 256 records at a time are put together in LDS and leave as whole-wavefront runs of consecutive
     * words: the destination is host memory, where a lane-strided struct store costs a PCIe write per
     * piece */
    __shared__ msd_wire sh_rec[256];
    __shared__ msd_fields sh_f[FIELDS ? 256 : 1];
    for (uint32_t m0 = 0; m0 < nm; m0 += 256) {
        const uint32_t m = m0 + threadIdx.x;
        if (m < nm) {
            unsigned long long sw;
            const msd_message mm = msd_emit_mode_s(acc[m], P.tries, power[(size_t)b * MSD_RB_MSG_CAP + m], sample_ts, sys_ts, base, sw);
            if (o + m < cap)
                side[o + m] = sw;
            sh_rec[threadIdx.x].mm = mm;
            if (FIELDS) /* MSD_CFG_DECODE_FIELDS: the header fields, from the corrected bytes */
                msd_fields_mode_s(mm.msg, mm.msgtype, mm.addr, &sh_f[threadIdx.x]);
        }
        __syncthreads();
        /* what does not fit is dropped: the host notices (total > cap), grows the arrays and emits again */
        emit_rows(dense, fields, sh_rec, sh_f, o + m0, min(256u, nm - m0), cap);
        __syncthreads();
    }
    if (P.ac) { /* the buffer's Mode A/C replies follow its Mode S messages (readsb.c:826-829) */
        const uint32_t na = P.nac[b];
        const uint32_t *acc_ac = P.acc_ac + (size_t)b * MSD_RB_AC_CAP;
        __shared__ uint32_t has_alt[MSD_RB_AC_CAP / 32]; /* replies that carry an altitude of their own */
        if (FIELDS) {
            for (uint32_t i = threadIdx.x; i < MSD_RB_AC_CAP / 32; i += blockDim.x)
                has_alt[i] = 0;
            __syncthreads();
            for (uint32_t m = threadIdx.x; m < na; m += blockDim.x) {
                const uint32_t a = P.ac[acc_ac[m]].modeac;
                if (!(a & 0x0080u) && msd_mode_a_to_c(a) != MSD_INVALID_ALTITUDE)
                    atomicOr(&has_alt[m >> 5], 1u << (m & 31));
            }
            __syncthreads();
        }
        for (uint32_t m0 = 0; m0 < na; m0 += 256) {
            const uint32_t m = m0 + threadIdx.x;
            if (m < na) {
                const msd_ac_hit c = P.ac[acc_ac[m]];
                const msd_message mm = msd_emit_mode_ac(c, sample_ts, sys_ts);
                if (o + nm + m < cap)
                    side[o + nm + m] = 0;
                sh_rec[threadIdx.x].mm = mm;
                if (FIELDS) {
                    /* the reference's one message record per buffer keeps the last decoded altitude
                     * (demod_2400.c:523-528): find the last earlier reply of this buffer that had one */
                    msd_fields carry;
                    const msd_fields *cp = nullptr;
                    int32_t w = (int32_t)(m >> 5);
                    uint32_t bits = m & 31 ? has_alt[w] & ((1u << (m & 31)) - 1u) : 0u;
                    while (!bits && --w >= 0)
                        bits = has_alt[w];
                    if (bits) {
                        const uint32_t prev = (uint32_t)w * 32 + (31u - (uint32_t)__builtin_clz(bits));
                        msd_fields_mode_ac(P.ac[acc_ac[prev]].modeac, nullptr, &carry);
                        cp = &carry;
                    }
                    msd_fields_mode_ac(c.modeac, cp, &sh_f[threadIdx.x]);
                }
            }
            __syncthreads();
            emit_rows(dense, fields, sh_rec, sh_f, o + nm + m0, min(256u, na - m0), cap);
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(64) msd_pred_patch_kernel(unsigned long long *table, const msd_pred_patch *patches, uint32_t n)
{
    /* the first buffer is the low half of the entry */
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        reinterpret_cast<uint32_t *>(table + patches[i].slot)[0] = patches[i].first;
}

/* The small per-batch results the host waits for -- list totals and per-buffer level/power sums --
 * written straight into pinned host memory at the end of the batch's kernels: a copy on another
 * stream would queue behind the persistent scan kernels of the following batches. */
__global__ void __launch_bounds__(256) msd_publish_kernel(const uint64_t *totals, const uint64_t *ac_totals,
                                                          uint64_t *sums, const float *fmeans, uint32_t nbuffers,
                                                          uint64_t *h_totals, uint64_t *h_ac_totals, uint64_t *h_sums,
                                                          float *h_fmeans)
{
    const uint32_t tid = threadIdx.x;
    if (tid < 4) {
        h_totals[tid] = totals[tid];
        if (ac_totals)
            h_ac_totals[tid] = ac_totals[tid];
    }
    for (uint32_t i = tid; i < 2 * nbuffers; i += blockDim.x) {
        h_sums[i] = sums[i];
        sums[i] = 0; /* ready for the slot's next batch */
        if (fmeans)
            h_fmeans[i] = fmeans[i];
    }
}

} /* namespace */

extern "C" int msd_launch_publish(const uint64_t *totals, const uint64_t *ac_totals, uint64_t *sums,
                                  const float *fmeans, uint32_t nbuffers, uint64_t *h_totals, uint64_t *h_ac_totals,
                                  uint64_t *h_sums, float *h_fmeans, hipStream_t stream)
{
    hipLaunchKernelGGL(msd_publish_kernel, dim3(1), dim3(256), 0, stream, totals, ac_totals, sums, fmeans, nbuffers,
                       h_totals, h_ac_totals, h_sums, h_fmeans);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_pred_patch(unsigned long long *table, const msd_pred_patch *patches, uint32_t n, hipStream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(msd_pred_patch_kernel, dim3(1), dim3(64), 0, stream, table, patches, n);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

/* msd_decode_fields_device: the field decoder alone, one thread per Mode S message */
__global__ void __launch_bounds__(256) msd_fields_kernel(const msd_message *in, msd_fields *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const msd_message mm = in[i];
    msd_fields f;
    if (mm.msgtype == 32)
        msd_fields_mode_ac(((uint32_t)mm.msg[0] << 8) | mm.msg[1], nullptr, &f);
    else
        msd_fields_mode_s(mm.msg, mm.msgtype, mm.addr, &f);
    out[i] = f;
}

extern "C" int msd_launch_fields(const msd_message *d_in, msd_fields *d_out, uint32_t n, hipStream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(msd_fields_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_in, d_out, n);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_resolve(const MsdResolveParams *p, uint32_t ntodo, hipStream_t stream)
{
    if (ntodo == 0)
        return 0;
    hipLaunchKernelGGL(msd_resolve_kernel, dim3(ntodo), dim3(RT), 0, stream, *p);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_emit(const MsdResolveParams *p, uint32_t nbuffers, const unsigned long long *power,
                               unsigned long long *side, msd_wire *dense, msd_fields *fields, uint32_t cap,
                               hipStream_t stream)
{
    if (nbuffers == 0)
        return 0;
    if (fields)
        hipLaunchKernelGGL(msd_emit_kernel<true>, dim3(nbuffers), dim3(256), 0, stream, *p, power, side, dense, fields, cap);
    else
        hipLaunchKernelGGL(msd_emit_kernel<false>, dim3(nbuffers), dim3(256), 0, stream, *p, power, side, dense, fields, cap);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
