/*
 * msd_kernels.hip -- CDNA4 (gfx950) kernels of the Mode S / Mode A/C candidate stage.
 *
 * msd_scan_kernel fuses, per tile of 1024 scan positions:
 *   IQ -> u16 magnitude            convert.c:63-111 (UC8 table), :215-253 / :332-370 (float)
 *   preamble pre-check + 3 tests   demod_2400.c:276-330
 *   5-phase PPM bit slicing        demod_2400.c:73-229 (closed form t = 95 + tp + 12k)
 *   CRC-24 + syndrome lookup       crc.c:67-82, :389-412
 *   state-free part of scoring     mode_s.c:311-409
 * Magnitudes live only in LDS: HBM sees each IQ byte once and a few bytes of candidate records
 * per thousand samples.  No MFMA: there is no dense contraction anywhere on this path.
 *
 * Work decomposition
 *   - wave-autonomous: one 1024-thread workgroup per CU only shares the constant tables (the UC8
 *     magnitude table above all); after they are in LDS the 16 wavefronts never meet again -- there
 *     is no workgroup barrier in the loop.  Every wavefront owns a contiguous run of 1024-position
 *     tiles (a "region"), a private 6.3 KB block of LDS and a private slice of the candidate arenas;
 *     the 328-sample look-behind of a tile is carried over inside LDS from the previous tile, and
 *     the next tile's IQ is prefetched into registers while the current one is processed;
 *   - the scan gives every lane 16 consecutive positions (register-tiled sliding window: 5 LDS
 *     reads of 16 B feed 16 positions x 19 taps);
 *   - the candidate stage works on a tile's hits in rounds of up to 64 (one per lane): the bit
 *     slicer is phase-agnostic -- 12 samples hold exactly five message bits, so a lane slices a
 *     "group" of five bits by evaluating each of the five correlators once, at a sample offset and
 *     into a bit position that depend on the trial phase only through two small tables -- which
 *     lets any lane take any (try, group) item; the CRC is put together from per-group syndrome
 *     tables (modesChecksum is linear) with LDS atomics;
 *   - hits and tries leave in position order through wave-uniform cursors -- no global atomics;
 *     msd_gather_kernel concatenates the regions into the dense, ordered lists the resolve stage walks.
 */
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "msd_internal.h"
#include "msd_dc_chain_asm.h"
#include "msd_kernels.h"

/* Candidate arenas (hits, tries): written once by the scan, read once by the next kernel.  MSD_ARENA_NT=1 marks the
 * stores non-temporal, so that the lines leave the L2 while the kernel runs instead of in its end-of-kernel write-back. */
#ifndef MSD_ARENA_NT
#define MSD_ARENA_NT 1
#endif
#if MSD_ARENA_NT
#define MSD_ARENA_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define MSD_ARENA_STORE(v, p) (*(p) = (v))
#endif
typedef uint32_t msd_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void arena_store16(const uint4 v, uint4 *p)
{
#if MSD_ARENA_NT
    msd_v4u x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<msd_v4u *>(p));
#else
    *p = v;
#endif
}
/* s_setprio per phase of a tile (0..3; the SIMD's issue arbitration takes the highest priority first, then the oldest
 * wavefront).  With all phases alike the four wavefronts of a SIMD are served by age, and a wavefront in its candidate
 * rounds -- short bursts of a few instructions between LDS round trips -- queues behind the long instruction runs of a
 * neighbour's preamble tests every time it comes back from a wait. */
#ifndef MSD_AC_PRIO
#define MSD_AC_PRIO 2 /* msd_ac_wave_kernel: the F2 test and the bit windows above the F1 test (145 -> 138 us) */
#endif
#ifndef MSD_SLICER_SIGN
#define MSD_SLICER_SIGN 1 /* the slicer's verdicts as sign bits (group_code): 0.2571 -> 0.2535 ms per 128 Mi samples */
#endif
#ifndef MSD_TESTS_SWZ
#define MSD_TESTS_SWZ 1 /* the preamble tests' 16-byte LDS reads in an order that uses every bank (stage 2): SQ_LDS_BANK_CONFLICT
                           27.9 M -> 20.9 M, SQ_LDS_IDX_ACTIVE 50.9 M -> 43.5 M cycles per 64 Mi-sample launch, the launch itself
                           0.2617 -> 0.2596 ms per 128 Mi samples (profiles/r04_pmc_counters.txt) */
#endif
#ifndef MSD_PRIO_CONV
#define MSD_PRIO_CONV 0
#endif
#ifndef MSD_TESTS_PRE_PLANE
#define MSD_TESTS_PRE_PLANE 1
#endif
#ifndef MSD_PRIO_GATHER
#define MSD_PRIO_GATHER MSD_PRIO_CONV /* the table gathers' addresses and issue (experiment: above the rest of the conversion) */
#endif
#ifndef MSD_PRIO_TESTS
#define MSD_PRIO_TESTS 1
#endif
#ifndef MSD_PRIO_CAND
#define MSD_PRIO_CAND 2 /* measured (profiles/r04_priorities.txt): 0/0/0 0.297 ms per 128 Mi samples, 0/1/3 0.266, 0/0/1..3 0.269-0.273,
                           0/1/2 0.267, 1/2/3 0.264, 1/1/3 0.278, 0/1/3 with step B at 0 / 1 / 2: 0.284 / 0.278 / 0.269 */
#endif
#ifndef MSD_PRIO_STEPB
#define MSD_PRIO_STEPB 3 /* step B of a candidate round: most of the round's LDS traffic */
#endif
#include "msd_emit_impl.h"
#include "msd_pred_impl.h"
#include "msd_mag_impl.h"

/* The float converters must round like the reference's x86-64 build: separate multiply and add
 * (no FMA contraction) and a correctly rounded square root.  The file is compiled with
 * -ffp-contract=off as well; sqrtf is IEEE-exact under hipcc's default
 * -fhip-fp32-correctly-rounded-divide-sqrt (the __f*_rn / __fsqrt_rn intrinsics are NOT: they map
 * to plain operators and the native sqrt). */
#pragma clang fp contract(off)

#ifdef MSD_KERNEL_TIMING
/* section timers for experiments (never in the shipped build): wall cycles per wavefront */
__device__ unsigned long long g_msd_tlast_dummy;
/* (gfx950 has no SHADER_CYCLES register: s_memtime it is -- a scalar memory operation whose s_waitcnt also waits for
 * every LDS access in flight, so the marks distort what they measure; keep them few) */
__device__ __forceinline__ uint32_t msd_cycles()
{
    return (uint32_t)__builtin_readcyclecounter();
}
#define TDECL uint32_t tacc_[12] = {0}; uint32_t tlast_ = msd_cycles();
#define TCOARSE(k) (MSD_KERNEL_TIMING + 0 != 2 || (k) == 0 || (k) == 2 || (k) == 9) /* -DMSD_KERNEL_TIMING=2: three marks per tile */
#define TMARK(k) if (TCOARSE(k)) { const uint32_t n_ = msd_cycles(); tacc_[k] += n_ - tlast_; tlast_ = n_; }
#define TMARKF(k) if (MSD_KERNEL_TIMING + 0 != 2) { const uint32_t n_ = msd_cycles(); tacc[k] += n_ - *tlast; *tlast = n_; }
#define TFLUSH if (P.timers && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&P.timers[k_], (unsigned long long)tacc_[k_]); }
#else
#define TDECL
#define TMARK(k)
#define TMARKF(k)
#define TFLUSH
#endif

namespace {

constexpr int WAVES = MSD_SCAN_WAVES; /* wavefronts per workgroup, one workgroup per CU */
constexpr int NT = MSD_SCAN_THREADS;
constexpr int WT_MAX = MSD_TILE;      /* scan positions per wavefront tile: 16 per lane and run, see tile_runs() */
constexpr int FRONT = MSD_HALO_FRONT; /* 328 samples of look-behind staged ahead of a tile */
/* runs of 16 consecutive positions per lane and tile: two for the byte formats; the 16-bit IQ formats hold
 * twice the raw data per sample in registers (current and prefetched tile) and stay at one (round 4: with two the SC16
 * scan spills ten registers and is 6 % faster, 0.186 -> 0.175 ms per 64 Mi samples -- and level again, 0.181 against
 * 0.184, once the tile's two 1024-sample blocks flush their float-sum predictions apart) */
__host__ __device__ constexpr int tile_runs(int fmt)
{
    return (fmt == MSD_FMT_SC16 || fmt == MSD_FMT_SC16Q11) ? 1 : WT_MAX / 1024;
}
constexpr int HC = 64;                /* hits per candidate round: one per lane */
constexpr int SC = 64;                /* tries with a known DF per round: one per lane in step C; a round
                                         that would need more is retried with half the hits */
constexpr int LUT_STRIDE = MSD_LUT_STRIDE;

static_assert(WT_MAX == 1024 || WT_MAX == 2048 || WT_MAX == 4096, "each lane scans one, two or four runs of 16 consecutive positions");
static_assert(WT_MAX <= 8192, "a position inside a tile has 13 bits in the hit and slot words");
static_assert(FRONT % 8 == 0, "whole load groups");
static_assert(MSD_CHUNK_SAMPLES % WT_MAX == 0, "a tile never straddles two buffers");

/* ---- dynamic LDS: tables shared by the workgroup, one private block per wavefront, the UC8 table ---- */
constexpr int OFF_SYN = 0;                                  /* u32[MSD_SYNH_WORDS]: the single-bit syndromes in buckets of four */
constexpr int OFF_SL = OFF_SYN + MSD_SYNH_WORDS * 4;        /* u32[MSD_SLICER_WORDS] */
constexpr int OFF_WGC = OFF_SL + MSD_SLICER_WORDS * 4;      /* u32[WAVES][4]: the regions' counts, at the end */
constexpr int OFF_WAVE = (OFF_WGC + WAVES * 16 + 15) & ~15;
constexpr int W_MAGS = 0;                                   /* u16[FRONT + WT_MAX + 8] */
constexpr int W_HITS = W_MAGS + (FRONT + WT_MAX + 8) * 2;   /* u32[HC]: position | tests << 13 */
constexpr int W_TRYL = W_HITS + HC * 4;                     /* u16[5 * HC]: hit | q << 8 */
constexpr int W_SIDX = W_TRYL + HC * 5 * 2;                 /* u16[HC][8]: slot of try (hit, q), 0xffff = none */
constexpr int W_SMETA = W_SIDX + HC * 8 * 2;                /* u32[SC] */
constexpr int W_SMSG = W_SMETA + SC * 4;                    /* uint4[SC] */
constexpr int W_SCRC = W_SMSG + SC * 16;                    /* u32[SC] */
constexpr int W_SRES = W_SCRC + SC * 4;                     /* u32[SC][2]: addr, crc */
constexpr int W_SQOFF = W_SRES + SC * 8;                    /* u32[SC]: MSD_SL_QOFF of the slot's trial phase */
constexpr int W_BYTES = W_SQOFF + SC * 4;
constexpr int OFF_LUT = OFF_WAVE + WAVES * W_BYTES;         /* u16[128 * LUT_STRIDE], UC8 only */
constexpr int LDS_COMMON = OFF_LUT;
constexpr int LDS_UC8 = OFF_LUT + (MSD_LUT_GLOBAL ? 0 : 128 * LUT_STRIDE * 2);
static_assert(OFF_WAVE % 16 == 0 && W_SMSG % 16 == 0 && W_BYTES % 16 == 0 && OFF_LUT % 16 == 0 && W_SIDX % 4 == 0,
              "LDS carve offsets must stay aligned");
static_assert(LDS_UC8 <= 160 * 1024, "one workgroup per CU: 160 KiB of LDS");


template <int FMT>
struct RawGroup { /* the raw bytes of 8 consecutive samples */
    static constexpr int WORDS = (FMT == MSD_FMT_SC16 || FMT == MSD_FMT_SC16Q11) ? 8 : 4;
    uint32_t w[WORDS];
};

/* Fetch the raw bytes of samples [n, n+8); n is a multiple of 8.  Returns a bit per sample that
 * exists; samples outside the stream (before its start / after a discontinuity / past the end)
 * have magnitude zero (fifo.c:179-182).
 * Always exactly one unconditional vector load from a *selected* address: a load inside a branch
 * forces the compiler to wait for it at the join (s_waitcnt vmcnt(0) right behind the load), which
 * would expose the full HBM latency of the next tile's prefetch.  A partially valid last group is
 * read from P.ragged, a 32-byte zero-padded copy prepared by the host; groups that do not exist
 * read the (always present) lookup table and are masked out. */
template <int FMT>
__device__ __forceinline__ uint32_t fetch_group(const MsdScanParams &P, int64_t n, RawGroup<FMT> &r)
{
    constexpr int BPS = RawGroup<FMT>::WORDS / 2;
    const int64_t rel = n - (int64_t)P.batch_first;
    const int64_t left = (int64_t)P.nsamples - rel;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(P.lut);
    uint32_t valid = 0;
    if (rel < 0) {
        if (P.have_prev && rel >= -(int64_t)FRONT) {
            src = P.prev_tail + (rel + FRONT) * BPS;
            valid = 0xffu;
        }
    } else if (left >= 8) {
        src = P.iq + rel * BPS;
        valid = 0xffu;
    } else if (left > 0) {
        src = P.ragged;
        valid = (1u << (int)left) - 1u;
    }
    const uint4 a = *reinterpret_cast<const uint4 *>(src);
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
    if (BPS == 4) {
        const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
        r.w[4 % RawGroup<FMT>::WORDS] = b.x; r.w[5 % RawGroup<FMT>::WORDS] = b.y;
        r.w[6 % RawGroup<FMT>::WORDS] = b.z; r.w[7 % RawGroup<FMT>::WORDS] = b.w;
    }
    return valid;
}

template <int FMT, bool MASK = true /* false: the caller masks (mask_group) once all its groups' loads are out */,
          bool SCAN_TABLE = false /* lut is the context's device table: use the 256-pitch copy behind it (msd_internal.h) */>
__device__ __forceinline__ void convert_group(const RawGroup<FMT> &r, uint32_t valid, const uint16_t *lut,
                                              uint32_t (&mg)[8])
{
    if (FMT == MSD_FMT_UC8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            /* fold8 of the four bytes I0 Q0 I1 Q1 at once: bytes with the top bit set keep their low
             * seven bits, the others are complemented */
            const uint32_t w = r.w[k];
            const uint32_t top = w & 0x80808080u;
            const uint32_t keep = top - (top >> 7); /* 0x7f in the bytes >= 128 */
            const uint32_t f = (w ^ ~keep) & 0x7f7f7f7fu;
            if constexpr (SCAN_TABLE) {
                /* the 256-pitch copy behind the folded table (msd_internal.h): the folded pair is the index.  Both samples'
                 * columns are swizzled by one XOR (no carries between the halves), both byte offsets come out of one
                 * addition (bit 15 of f is zero, so f + f carries nothing into the second sample's field), one AND and
                 * one shift -- all of them full-rate forms, written as asm so that they stay that way. */
                const unsigned char *tab = reinterpret_cast<const unsigned char *>(lut + MSD_LUT_SCAN_OFFSET);
                uint32_t x = f, t, o0, o1;
                if (MSD_LUT_SCAN_SWZ_BITS) {
                    constexpr uint32_t SM = ((1u << MSD_LUT_SCAN_SWZ_BITS) - 1u) << MSD_LUT_SCAN_SWZ_SHIFT;
                    uint32_t r;
                    asm("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "n"(8 - MSD_LUT_SCAN_SWZ_SHIFT), "v"(f));
                    asm("v_and_b32 %0, %1, %2" : "=v"(r) : "n"(SM | (SM << 16)), "v"(r));
                    asm("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(f), "v"(r));
                }
                asm("v_add_u32 %0, %1, %1" : "=v"(t) : "v"(x));
                asm("v_and_b32 %0, 0x1fffe, %1" : "=v"(o0) : "v"(t));
                asm("v_lshrrev_b32 %0, 16, %1" : "=v"(o1) : "v"(t));
                mg[2 * k] = *reinterpret_cast<const uint16_t *>(tab + o0);
                mg[2 * k + 1] = *reinterpret_cast<const uint16_t *>(tab + o1);
                continue;
            }
            mg[2 * k] = lut[((f >> 8) & 0xffu) * LUT_STRIDE + (f & 0xffu)];
            mg[2 * k + 1] = lut[(f >> 24) * LUT_STRIDE + ((f >> 16) & 0xffu)];
        }
    } else if (FMT == MSD_FMT_MAG16) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            mg[k] = (r.w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
    } else {
        const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t w = r.w[k % RawGroup<FMT>::WORDS];
            mg[k] = mag_from_s16((int)(int16_t)(w & 0xffffu), (int)(int16_t)(w >> 16), inv);
        }
    }
    if (MASK && valid != 0xffu) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (!((valid >> k) & 1u))
                mg[k] = 0;
    }
}

/* samples of the group that do not exist (before the stream, behind a gap, past the end) are silence (fifo.c:179-182) */
__device__ __forceinline__ void mask_group(uint32_t valid, uint32_t (&mg)[8])
{
    if (valid != 0xffu) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (!((valid >> k) & 1u))
                mg[k] = 0;
    }
}

__device__ __forceinline__ uint4 pack8(const uint32_t (&mg)[8])
{
    uint4 p;
    p.x = mg[0] | (mg[1] << 16);
    p.y = mg[2] | (mg[3] << 16);
    p.z = mg[4] | (mg[5] << 16);
    p.w = mg[6] | (mg[7] << 16);
    return p;
}


/* both halves of a sample pair shifted right by five (v_pk_lshrrev_b16) */
__device__ __forceinline__ uint32_t pk_shr5(uint32_t pair)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, pair) >> (unsigned short)5);
}

/* the exact sum of squares from its value modulo 2^32 and the bracketing sum of the truncated squares
 * (see msd_scan_kernel): the one number congruent to `mod` in [1024 * top, 1024 * top + 2^32) */
__device__ __forceinline__ unsigned long long power_sum(uint32_t mod, uint32_t top)
{
    const unsigned long long low = (unsigned long long)top << 10;
    return low + (uint32_t)(mod - (uint32_t)low);
}

__device__ __forceinline__ uint32_t dot2u(uint32_t pair, uint32_t weights, uint32_t acc)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pair), __builtin_bit_cast(us2, weights), acc, false);
}

/* demod_2400.c:193-205 */
__device__ __forceinline__ uint32_t bytes_for_df(uint32_t df)
{
    /* short: 0,4,5,11  long: 16,17,18,20,21,24  else give up after the DF */
    const uint32_t short_set = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
    const uint32_t long_set = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21) | (1u << 24);
    if ((short_set >> df) & 1u)
        return 7;
    if ((long_set >> df) & 1u)
        return 14;
    return 1;
}

/* Inclusive prefix sum over the 64 lanes with DPP moves only (no LDS round trips): four shifted adds
 * inside each row of 16, then the last lane of row 0 / 2 into rows 1 / 3 and lane 31 into rows 2, 3. */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1, 3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2, 3 */
    return x;
}

__device__ __forceinline__ uint32_t wave_last(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* order LDS traffic between lanes of one wavefront (they run in lock step and the LDS executes a
 * wavefront's accesses in order; this only stops the compiler from moving accesses across the
 * exchange point) */
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* `plane = 2 * plane + verdict` for every lane in one v_addc_co_u32: the verdicts of the wavefront are a
 * lane mask in scalar registers (v_cmp), used as the carry-in */
#define MSD_PUSH(PLANE, MASK) asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(PLANE) : "s"(MASK) : "vcc")

/* The five correlators of demod_2400.c:73-93 on one 5-bit group: correlator c looks at three (c == 4:
 * four) consecutive samples from LDS byte address a[c] + EXTRA; is its bit a one?  (`18 m0 - 15 m1 -
 * 3 m2 > 0` and so on, with the negative terms moved to the other side.)  Returns the five verdicts,
 * correlator 0 in bit 4; EXTRA = 24 bytes per further group of the same try (12 samples hold exactly
 * five bits, so every group repeats the same five (correlator, offset) pairs).
 * The samples are read one by one on purpose: the addresses are only 2-byte aligned, and a misaligned
 * ds_read_b32/b64 -- which the compiler would merge them into -- is replayed lane by lane on gfx950
 * (measured: 9x).  All sixteen loads are issued before the first verdict is worked out, so a group
 * costs one LDS latency, not five.  (Loading the samples straight into the halves of packed pairs for
 * v_dot2_u32_u16 does not work here: with SRAM ECC a ds_read_u16_d16_hi zeroes the other half of its
 * destination instead of keeping it.) */
template <int EXTRA>
__device__ __forceinline__ void group_load(const unsigned char *const (&a)[5], uint32_t (&m)[5][4])
{
    /* (an explicit LDS pointer: the compiler does not infer the address space of a volatile access and
     * would emit flat loads) */
    typedef __attribute__((address_space(3))) const volatile uint16_t lds_sample;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        lds_sample *p = (lds_sample *)(a[c] + EXTRA);
        m[c][0] = p[0];
        m[c][1] = p[1];
        m[c][2] = p[2];
        m[c][3] = c == 4 ? p[3] : 0u;
    }
}

/* The five verdicts of a group from its samples, as the group's five bits under trial phase 4 (q = 0): there bit k is
 * correlator (4 + 2 k) % 5 = 4, 1, 3, 0, 2, and under trial phase 4 + q every correlator's bit moves 2 q places on
 * (mod 5) -- the group's bits are this code rotated right by (2 q) % 5 within its five bits (group_bits). */
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ uint32_t group_code(const uint32_t (&m)[5][4])
{
    uint32_t code = 0;
#if MSD_SLICER_SIGN
    /* every verdict as the sign bit of the NEGATED correlator sum (a one iff the sum is positive), put together with
     * v_mad_i32_i24 (samples < 2^16, weights <= 20: no overflow) and pushed with one v_alignbit -- instead of products, a
     * compare into a lane mask and an add-with-carry.  Correlator 0's weights 18, -15, -3 are 3 x (6, -5, -1). */
    const int c6 = -6, c5 = 5, c14 = -14, c16 = -16, cm5 = -5, c20 = 20, c7 = -7, c11 = -11, c18 = 18, c4 = -4, c15 = -15;
    const int y4 = mad24((int)m[4][2], c20, mad24((int)m[4][1], c15, mad24((int)m[4][0], c4, -(int)m[4][3])));
    code = __builtin_amdgcn_alignbit(code, (uint32_t)y4, 31);
    const int y1 = mad24((int)m[1][0], c14, mad24((int)m[1][1], c5, (int)(m[1][2] * 9u)));
    code = __builtin_amdgcn_alignbit(code, (uint32_t)y1, 31);
    const int y3 = mad24((int)m[3][2], c18, mad24((int)m[3][1], c11, (int)m[3][0] * c7));
    code = __builtin_amdgcn_alignbit(code, (uint32_t)y3, 31);
    const int y0 = mad24((int)m[0][1], c5, mad24((int)m[0][0], c6, (int)m[0][2]));
    code = __builtin_amdgcn_alignbit(code, (uint32_t)y0, 31);
    const int y2 = mad24((int)m[2][2], c20, mad24((int)m[2][1], cm5, (int)m[2][0] * c16));
    code = __builtin_amdgcn_alignbit(code, (uint32_t)y2, 31);
#else
    const uint64_t b4 = __ballot(4u * m[4][0] + 15u * m[4][1] + m[4][3] > 20u * m[4][2]);
    MSD_PUSH(code, b4);
    const uint64_t b1 = __ballot(14u * m[1][0] > 5u * m[1][1] + 9u * m[1][2]);
    MSD_PUSH(code, b1);
    const uint64_t b3 = __ballot(7u * m[3][0] + 11u * m[3][1] > 18u * m[3][2]);
    MSD_PUSH(code, b3);
    const uint64_t b0 = __ballot(18u * m[0][0] > 15u * m[0][1] + 3u * m[0][2]);
    MSD_PUSH(code, b0);
    const uint64_t b2 = __ballot(16u * m[2][0] + 5u * m[2][1] > 20u * m[2][2]);
    MSD_PUSH(code, b2);
#endif
    return code;
}

/* (2 q) % 5 for q = 0..4: how far the bits of a group move between trial phase 4 and trial phase 4 + q */
__device__ __forceinline__ uint32_t phase_rot(uint32_t q)
{
    return (0x31420u >> (4u * q)) & 7u;
}

/* the group's five message bits (first bit in bit 4) from group_code and phase_rot(q) */
__device__ __forceinline__ uint32_t group_bits(uint32_t code, uint32_t rot)
{
    return ((code | (code << 5)) >> rot) & 31u;
}

template <int EXTRA>
__device__ __forceinline__ uint32_t group_verdicts(const unsigned char *const (&a)[5], uint32_t rot)
{
    uint32_t m[5][4];
    group_load<EXTRA>(a, m);
    return group_bits(group_code(m), rot);
}

/* Everything a wavefront needs besides its parameters: its private LDS block and the shared tables. */
struct WaveCtx {
    unsigned char *w;          /* the wavefront's private LDS block (W_* offsets) */
    const uint32_t *syn;       /* sorted single-bit syndrome tables */
    const uint32_t *sl;        /* slicer / CRC tables (MSD_SL_*) */
    int lane;
};

/* The candidate stage for one round of up to HC hits of a tile (hitl[0..nh), in position order, one per
 * lane).  Entirely wave-synchronous: no workgroup barrier, every exchange goes through the wavefront's
 * own LDS block.
 *
 *   tries   every hit expands into the trial phases its preamble tests ask for (demod_2400.c:298-330)
 *   step A  the first five bits of every try = the DF -> message length (demod_2400.c:183-205);
 *           tries with a known DF get a slot: short ones from slot 0 up, long ones from slot SC-1 down
 *   step B  the rest of the message, one lane per (slot, pair of 5-bit groups); every item also
 *           looks up its groups' share of the syndrome (modesChecksum is linear) -- LDS atomics
 *           put the message bits and the CRC together (crc.c:67-82)
 *   step C  syndrome lookup and the filter-independent part of scoreModesMessage, one lane per slot
 *           (crc.c:389-412; mode_s.c:311-409)
 *   step D  records out: per hit, its live tries in phase order at consecutive indices.
 * Returns false if the round has more than SC tries with a known DF (the caller halves it). */
template <bool FIX2>
__device__ __forceinline__ bool candidate_round(const MsdScanParams &P, const WaveCtx &X, uint32_t nh, uint64_t tile_pos0,
                                                msd_hit *hit_out, uint32_t hits_room, msd_try *my_tries,
                                                uint32_t &tcur, uint32_t try_base /* of my_tries[0] in the hit records */
#ifdef MSD_KERNEL_TIMING
                                                , uint32_t *tacc, uint32_t *tlast
#endif
                                                )
{
    const int lane = X.lane;
    const unsigned char *mbytes = X.w + W_MAGS;
    const uint32_t *hitl = reinterpret_cast<const uint32_t *>(X.w + W_HITS);
    uint16_t *tryl = reinterpret_cast<uint16_t *>(X.w + W_TRYL);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(X.w + W_SIDX);
    uint32_t *smeta = reinterpret_cast<uint32_t *>(X.w + W_SMETA);
    uint32_t *smsg32 = reinterpret_cast<uint32_t *>(X.w + W_SMSG);
    uint32_t *scrc = reinterpret_cast<uint32_t *>(X.w + W_SCRC);
    uint32_t *sres = reinterpret_cast<uint32_t *>(X.w + W_SRES);
    uint32_t *sqoff = reinterpret_cast<uint32_t *>(X.w + W_SQOFF);

    /* ---- tries: lane h expands hit h ---- */
    uint32_t my_hit = 0, my_m = 0;
    {
        *reinterpret_cast<uint4 *>(sidx + 8 * lane) = make_uint4(~0u, ~0u, ~0u, ~0u);
        if ((uint32_t)lane < nh) {
            my_hit = hitl[lane];
            my_m = my_hit >> 13;
        }
    }
    const uint32_t nt = 2u * (my_m & 1u) + (my_m & 2u) + ((my_m >> 2) & 1u);
    const uint32_t tincl = wave_incl_scan(nt);
    const uint32_t ntry = wave_last(tincl);
    {
        uint32_t k = tincl - nt;
        if (my_m & 1u) {
            tryl[k] = (uint16_t)lane;
            tryl[k + 1] = (uint16_t)(lane | (1 << 8));
            k += 2;
        }
        if (my_m & 2u) {
            tryl[k] = (uint16_t)(lane | (2 << 8));
            tryl[k + 1] = (uint16_t)(lane | (3 << 8));
            k += 2;
        }
        if (my_m & 4u)
            tryl[k] = (uint16_t)(lane | (4 << 8));
    }
    wave_lds_sync();

    /* ---- step A ---- */
    uint32_t ns = 0, nl = 0; /* wave-uniform: short / long slots handed out */
    for (uint32_t i0 = 0; i0 < ntry; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool act = i < ntry;
        const uint32_t e = act ? tryl[i] : 0u;
        const uint32_t h = e & 0xffu, q = e >> 8;
        const uint32_t pos = hitl[h] & 0x1fffu;
        const uint32_t qoff = X.sl[MSD_SL_QOFF + q];
        /* pa[0] = mags[pos + 2]: the tile stages 328 samples ahead, the reference's overlap is 326 */
        const unsigned char *base = mbytes + 2u * pos + 4u;
        const unsigned char *a[5];
#pragma unroll
        for (int c = 0; c < 5; ++c)
            a[c] = base + ((qoff >> (6 * c)) & 63u);
        const uint32_t df = group_verdicts<0>(a, phase_rot(q));
        const uint32_t nb = bytes_for_df(df);
        const bool is_s = act && nb == 7, is_l = act && nb == 14;
        const uint64_t bs = __ballot(is_s), bl = __ballot(is_l);
        const uint32_t rs = __builtin_amdgcn_mbcnt_hi((uint32_t)(bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs, 0u));
        const uint32_t rl = __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
        const uint32_t u = is_s ? ns + rs : (uint32_t)(SC - 1) - (nl + rl);
        ns += (uint32_t)__popcll(bs);
        nl += (uint32_t)__popcll(bl);
        if (ns + nl > (uint32_t)SC)
            return false; /* wave-uniform */
        if (is_s || is_l) {
            smeta[u] = pos | (q << 13) | (nb << 16) | (h << 20);
            *reinterpret_cast<uint4 *>(smsg32 + 4u * u) = make_uint4(df << 27, 0u, 0u, 0u);
            scrc[u] = X.sl[(is_l ? MSD_SL_GLONG : MSD_SL_GSHORT) + df];
            sqoff[u] = qoff;
            sidx[h * 8u + q] = (uint16_t)u;
        }
    }
    wave_lds_sync();
    TMARKF(4)

    /* ---- step B: item = (slot, NG consecutive groups from NG it + 1 on).  Groups 1..11 carry bits 5..55 of
     * a short message (the last one one bit), groups 1..22 bits 5..111 of a long one (the last one two).
     * NG groups = 24 NG bytes of samples per item: with NG = 3 the items of a try start 18 LDS banks apart,
     * all on different banks (with four groups per item, 24 banks apart, every try collided with itself) ---- */
    if (!(P.debug_flags & 4)) {
        if (MSD_PRIO_STEPB != MSD_PRIO_CAND)
            __builtin_amdgcn_s_setprio(MSD_PRIO_STEPB);
        constexpr uint32_t NG = MSD_SLICER_NG, IS = (11 + NG - 1) / NG, IL = (22 + NG - 1) / NG;
        constexpr uint32_t VB = 5 * NG; /* bits per item */
        constexpr uint32_t LAST_S = 51 - VB * (IS - 1), LAST_L = 107 - VB * (IL - 1); /* bits of the last item */
        constexpr uint32_t MASK_S = ((1u << LAST_S) - 1u) << (VB - LAST_S), MASK_L = ((1u << LAST_L) - 1u) << (VB - LAST_L);
        const uint32_t short_items = IS * ns, nitems = short_items + IL * nl;
        for (uint32_t i0 = 0; i0 < nitems; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool act = i < nitems;
            const bool lng = i >= short_items;
            const uint32_t j = act ? (lng ? i - short_items : i) : 0u;
            const uint32_t t = (j * (lng ? (65536u + IL - 1) / IL : (65536u + IS - 1) / IS)) >> 16; /* j / IL or j / IS (j < 1024) */
            const uint32_t it = j - t * (lng ? IL : IS);
            const uint32_t u = lng ? (uint32_t)(SC - 1) - t : t;
            const uint32_t me = smeta[u];
            const uint32_t qoff = sqoff[u]; /* (with the slot's record: one LDS round trip, not two) */
            const uint32_t pos = me & (uint32_t)(WT_MAX - 1), q = (me >> 13) & 7u;
            const uint32_t g1 = NG * it + 1u;
            const unsigned char *base = mbytes + 2u * pos + 4u + 24u * g1;
            const unsigned char *a[5];
#pragma unroll
            for (int c = 0; c < 5; ++c)
                a[c] = base + ((qoff >> (6 * c)) & 63u);
            const uint32_t rot = phase_rot(q);
            uint32_t v[NG];
            v[0] = group_verdicts<0>(a, rot);
            if (NG > 1) v[1 % NG] = group_verdicts<24>(a, rot);
            if (NG > 2) v[2 % NG] = group_verdicts<48>(a, rot);
            if (NG > 3) v[3 % NG] = group_verdicts<72>(a, rot);
            if (NG > 4) v[4 % NG] = group_verdicts<96>(a, rot);
            uint32_t val = 0; /* message bits 5 g1 .. 5 g1 + VB - 1 */
#pragma unroll
            for (uint32_t c = 0; c < NG; ++c)
                val |= v[c] << (5u * (NG - 1u - c));
            if (it == (lng ? IL - 1u : IS - 1u))
                val &= lng ? MASK_L : MASK_S; /* bits past the end of the message */
            if (act) {
                const uint32_t *gt = X.sl + (lng ? MSD_SL_GLONG : MSD_SL_GSHORT) + 32u * g1;
                uint32_t syn = 0;
#pragma unroll
                for (uint32_t c = 0; c < NG; ++c) {
                    const uint32_t e = (val >> (5u * (NG - 1u - c))) & 31u;
                    if (NG * IL <= 24u && NG * IS <= 12u)
                        syn ^= gt[32u * c + e];
                    else if (e) /* rows past the last group do not exist */
                        syn ^= gt[32u * c + e];
                }
                atomicXor(&scrc[u], syn);
                /* message bit n lives in bit 31 - (n & 31) of word n >> 5 until step C */
                const uint32_t n0 = 5u * g1, s = n0 & 31u, top = val << (32u - VB);
                atomicOr(&smsg32[4u * u + (n0 >> 5)], top >> s);
                if (s > 32u - VB)
                    atomicOr(&smsg32[4u * u + (n0 >> 5) + 1u], __builtin_amdgcn_alignbit(top, 0u, s)); /* top << (32 - s) */
            }
        }
        if (MSD_PRIO_STEPB != MSD_PRIO_CAND)
            __builtin_amdgcn_s_setprio(MSD_PRIO_CAND);
    }
    wave_lds_sync();
    TMARKF(5)

    /* ---- step C: lane = slot ---- */
    if ((uint32_t)lane < ns || (uint32_t)lane >= (uint32_t)SC - nl) {
        const uint32_t u = (uint32_t)lane;
        const uint32_t me = smeta[u];
        const uint4 m4 = *reinterpret_cast<const uint4 *>(smsg32 + 4u * u);
        const uint32_t crc = scrc[u];
        const uint32_t df = m4.x >> 27, aa = m4.x & 0xffffffu;
        bool alive = (m4.x | m4.y | m4.z | m4.w) != 0; /* mode_s.c:325 */
        uint32_t addr = crc, errbit = 0xffu, errbit2 = 0xffu;
        if (alive && (df == 11 || df == 17 || df == 18)) {
            addr = aa;
            const uint32_t syndrome = (df == 11) ? (crc & 0xffff80u) : crc;
            if (syndrome != 0) {
                alive = false;
                if (FIX2) {
                    /* --aggressive: modesChecksumDiagnose against the (2, 4) tables, a hash probe in
                     * global memory (10 / 82 KiB, L2-resident) */
                    const uint64_t *tab = (df == 11) ? P.fix2_56 : P.fix2_112;
                    const uint32_t lg = (df == 11) ? P.fix2_lg56 : P.fix2_lg112;
                    uint32_t slot = MSD_FIX2_HASH(syndrome, lg);
                    for (;;) {
                        const uint64_t e = tab[slot];
                        if (e == ~0ull)
                            break;
                        if (((uint32_t)e & 0xffffffu) == syndrome) {
                            errbit = (uint32_t)(e >> 32) & 0xffu;
                            errbit2 = (uint32_t)(e >> 40) & 0xffu;
                            /* two wrong bits in a DF11 are ambiguous: never corrected (mode_s.c:352-356) */
                            alive = !(df == 11 && errbit2 != 0xffu);
                            break;
                        }
                        slot = (slot + 1) & ((1u << lg) - 1u);
                    }
                } else {
                    /* modesChecksumDiagnose (crc.c:389-412): exact match in the single-bit table, or give up.  The
                     * table lies in buckets of four (msd_internal.h): one 16-byte read, four compares, no loop. */
                    if (P.nsyn112) { /* wave-uniform: --no-fix has no table */
                        const uint32_t bkt = (df == 11) ? (syndrome * P.synh_mul56) >> (32u - MSD_SYNH_LG56)
                                                        : (4u << MSD_SYNH_LG56) / 4u + ((syndrome * P.synh_mul112) >> (32u - MSD_SYNH_LG112));
                        const uint4 e4 = *reinterpret_cast<const uint4 *>(X.syn + 4u * bkt);
                        const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((e[j] & 0xffffffu) == syndrome) {
                                errbit = e[j] >> 24;
                                alive = true;
                            }
                    }
                }
                if (alive && errbit >= 8 && errbit <= 31)
                    addr ^= 1u << (31 - errbit); /* correct_aa_field, mode_s.c:266-281 */
                if (alive && errbit2 >= 8 && errbit2 <= 31)
                    addr ^= 1u << (31 - errbit2);
            }
        }
        const uint32_t q = (me >> 13) & 7u, h = me >> 20;
        if (alive && crc == 0 && (df == 17 || df == 11) && P.pred) /* a clean DF17 / DF11 with II = 0: it will reach
                                                                     icaoFilterAdd if it is accepted (mode_s.c:717-726) */
            msd_pred_note(P.pred, P.pred_gen, aa, (uint32_t)(tile_pos0 / MSD_CHUNK_SAMPLES));
        if (alive) {
            /* the record holds the message as bytes, first byte first */
            *reinterpret_cast<uint4 *>(smsg32 + 4u * u) =
                make_uint4(__builtin_bswap32(m4.x), __builtin_bswap32(m4.y), __builtin_bswap32(m4.z),
                           (__builtin_bswap32(m4.w) & 0xffffu) | ((4u + q) << 16) | (errbit << 24));
            sres[2 * u] = addr;
            sres[2 * u + 1] = crc | (errbit2 << 24); /* the CRC has 24 bits */
        } else {
            sidx[h * 8u + q] = 0xffffu; /* scores -2 whatever the filter holds */
        }
    }
    wave_lds_sync();
    TMARKF(6)

    /* ---- step D: lane = hit ---- */
    uint32_t u5[5], nlive = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        u5[q] = (uint32_t)lane < nh ? (uint32_t)sidx[lane * 8 + q] : 0xffffu;
        nlive += u5[q] != 0xffffu ? 1u : 0u;
    }
    const uint32_t lincl = wave_incl_scan(nlive);
    const uint32_t ntry_round = wave_last(lincl);
    if ((uint32_t)lane < nh && (uint32_t)lane < hits_room) {
        const uint32_t pos = my_hit & 0x1fffu;
        msd_hit rec = (tile_pos0 + pos) | ((msd_hit)my_m << 28) | ((msd_hit)nlive << 31);
        if (nlive)
            rec |= (msd_hit)(try_base + tcur + lincl - nlive) << 34;
        hit_out[lane] = rec; /* 8 bytes a lane, a few lanes a round: left to the L2 to merge (non-temporal, every round's ragged ends went to memory on their own) */
    }
    /* The round's tries in output order (slot of the candidate scratch | owning lane << 9), in the index array every
     * lane has just read its row of; then two lanes per record write its two 16-byte halves, so that one store
     * instruction covers 1 KB of consecutive addresses -- whole lines, which is what a non-temporal store wants (a
     * lane per record, half by half, sent every 32-byte sector to memory twice). */
    wave_lds_sync();
    {
        uint32_t k = lincl - nlive;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (u5[q] != 0xffffu)
                sidx[k++] = (uint16_t)(u5[q] | ((uint32_t)lane << 9));
    }
    wave_lds_sync();
    for (uint32_t e0 = 0; e0 < ntry_round; e0 += 32u) { /* wave-uniform */
        const uint32_t e = e0 + ((uint32_t)lane >> 1);
        const bool mine = e < ntry_round && tcur + e < P.tcap;
        const uint32_t ent = mine ? (uint32_t)sidx[e] : 0u;
        const uint32_t u = ent & 0x1ffu;
        const uint32_t pos = (uint32_t)__shfl((int)(my_hit & 0x1fffu), (int)(ent >> 9), 64);
        if (mine) {
            uint4 *dst = reinterpret_cast<uint4 *>(my_tries + tcur + e);
            const uint4 m16 = *reinterpret_cast<const uint4 *>(smsg32 + 4u * u);
            if (!(lane & 1)) {
                arena_store16(m16, &dst[0]);
            } else {
                const uint32_t cw = sres[2 * u + 1];
                /* the last word repeats the first byte (DF), the trial phase and the first corrected bit beside the
                 * second one: the resolve kernel reads this half of the record only */
                arena_store16(make_uint4(sres[2 * u], cw & 0xffffffu, (uint32_t)(tile_pos0 + pos),
                                         (cw >> 24) | ((m16.x & 0xffu) << 8) | (m16.w & 0xffff0000u)),
                              &dst[1]);
            }
        }
    }
    tcur += ntry_round;
    wave_lds_sync();
    TMARKF(7)
    return true;
}

template <int FMT>
__device__ inline void msd_emit_slice(const MsdEmitJob &J, const uint16_t *lut_g, uint32_t w, int lane, unsigned char *lds,
                                      bool dbg_no_store = false);

/* One wavefront's share of the batch: the tiles [tile_lo, tile_hi) of 1024 * tile_runs(FMT) scan positions each. */
template <int FMT, bool FIX2, bool EMIT>
__device__ __forceinline__ void scan_region(const MsdScanParams &P, const WaveCtx &X, const uint16_t *lut,
                                            uint32_t region, uint32_t tile_lo, uint32_t tile_hi, uint32_t &hits_total,
                                            uint32_t &tries_total)
{
    constexpr int NH = tile_runs(FMT), WT = 1024 * NH; /* scan positions per tile */
    constexpr bool SCAN_LUT = MSD_LUT_SCAN256 && MSD_LUT_GLOBAL; /* lut is P.lut: the 256-pitch copy lies behind it */
    constexpr int GPT = WT / 8 / 64;                   /* 8-sample load groups per lane per tile (2 per run) */
    const int lane = X.lane;
    uint16_t *mags = reinterpret_cast<uint16_t *>(X.w + W_MAGS);
    uint32_t *hitl = reinterpret_cast<uint32_t *>(X.w + W_HITS);
    const uint64_t batch_end = P.batch_first + P.nsamples; /* one past the last scan position */
    msd_hit *const my_hits = P.hits + (size_t)region * P.hcap;
    msd_try *const my_tries = P.tries + (size_t)region * P.tcap;
    const uint32_t try_base = P.regions_per_buffer ? region * P.tcap : 0u; /* lean layout: arena-absolute try indices */
    uint32_t hcur = 0, tcur = 0; /* wave-uniform cursors into the region's slices of the arenas */

    /* running buffer sums (convert.c:78-110), flushed when the wavefront moves to another buffer */
    uint32_t sum_level = 0;
    /* Sum of the squares of at most 256 magnitudes per lane (16 tiles x 16 samples) without 64-bit
     * arithmetic in the loop: pw_mod = the sum modulo 2^32 (v_dot2_u32_u16 of a sample pair with itself
     * wraps), pw_top = the sum of (m >> 5)^2, which fits (2047^2 * 256 < 2^32) and brackets the true
     * sum: 1024 * pw_top <= sum < 1024 * pw_top + 256 * (64 * 2047 * 31 + 31^2) < 1024 * pw_top + 2^31.
     * power_sum() puts the two together when the buffer changes or after 256 samples per lane. */
    uint32_t pw_mod = 0, pw_top = 0, sum_tiles = 0, sum_tile0 = tile_lo; /* sum_tile0: first tile in the running sums */
    uint64_t sum_chunk = (uint64_t)tile_lo * WT / MSD_CHUNK_SAMPLES;
    /* 16-bit IQ (P.tile_sums): the sums of every single tile as floats (magnitude / 65535, its square) -- what
     * the float-sum kernel needs to predict the binade of its sequential sums at every 1024-sample block.  A
     * lane then holds 16 samples: its sums fit 20 and 36 bits, the wavefront's are three DPP scans, and the
     * buffer's exact sums wait in scalar registers until the buffer changes. */
    unsigned long long acc_level = 0, acc_power = 0;
    auto flush_sums = [&]() {
        if (P.chunk_sums && P.tile_sums) {
            if (sum_tiles) {
                const unsigned long long sp = power_sum(pw_mod, pw_top);
                const uint32_t tl = wave_last(wave_incl_scan((uint32_t)sum_level));
                const uint32_t tlo = wave_last(wave_incl_scan((uint32_t)sp & 0xffffffu));
                const uint32_t thi = wave_last(wave_incl_scan((uint32_t)(sp >> 24)));
                const unsigned long long tp = ((unsigned long long)thi << 24) + tlo;
                acc_level += tl;
                acc_power += tp;
                if (lane == 0) {
                    P.tile_sums[2 * sum_tile0] = (float)tl * (1.0f / 65535.0f);
                    P.tile_sums[2 * sum_tile0 + 1] = (float)tp * (1.0f / (65535.0f * 65535.0f));
                }
            }
        } else if (P.chunk_sums) {
            /* at most 256 samples per lane: the level sums fit 24 bits, the power sums 41 -- three DPP ladders instead of
             * two 64-bit shuffle ladders through the LDS */
            const unsigned long long spl = power_sum(pw_mod, pw_top);
            const unsigned long long sl = wave_last(wave_incl_scan((uint32_t)sum_level));
            const unsigned long long sp = (unsigned long long)wave_last(wave_incl_scan((uint32_t)spl & 0xffffffu)) +
                                          ((unsigned long long)wave_last(wave_incl_scan((uint32_t)(spl >> 24))) << 24);
            if (lane == 0 && (sl | sp)) {
                atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * sum_chunk]), sl);
                atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * sum_chunk + 1]), sp);
            }
        }
        sum_level = 0;
        pw_mod = 0;
        pw_top = 0;
        sum_tiles = 0;
    };
    auto flush_chunk = [&]() { /* tile mode: the buffer's exact sums, once per buffer and wavefront */
        if (P.chunk_sums && P.tile_sums && lane == 0 && (acc_level | acc_power)) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * sum_chunk]), acc_level);
            atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * sum_chunk + 1]), acc_power);
        }
        acc_level = 0;
        acc_power = 0;
    };

    /* look-behind of the first tile: samples [a0 - 328, a0) */
    {
        const uint64_t a0 = P.batch_first + (uint64_t)tile_lo * WT;
        if (lane < FRONT / 8) {
            RawGroup<FMT> r;
            const uint32_t valid = fetch_group<FMT>(P, (int64_t)a0 - FRONT + 8 * lane, r);
            uint32_t mg[8];
            convert_group<FMT, true, SCAN_LUT>(r, valid, lut, mg);
            *reinterpret_cast<uint4 *>(mags + 8 * lane) = pack8(mg);
        }
    }
    RawGroup<FMT> cur[GPT], nxt[GPT];
    uint32_t cur_valid[GPT], nxt_valid[GPT];
    {
        const uint64_t a0 = P.batch_first + (uint64_t)tile_lo * WT;
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            cur_valid[k] = fetch_group<FMT>(P, (int64_t)a0 + 8 * (lane + 64 * k), cur[k]);
            nxt_valid[k] = 0;
            nxt[k] = cur[k];
        }
    }
    /* the raw IQ of tile `ti` (which exists) into nxt: one unconditional load per group from a selected address (see
     * fetch_group); tiles that lie wholly inside the batch -- all but the last -- take the short way */
    auto fetch_tile = [&](uint32_t ti) {
        const int64_t rel = (int64_t)((uint64_t)ti * WT);
        if ((uint64_t)rel + WT <= (P.nsamples & ~7ull)) { /* wave-uniform */
            constexpr int BPS = RawGroup<FMT>::WORDS / 2;
#pragma unroll
            for (int k = 0; k < GPT; ++k) {
                const uint8_t *src = P.iq + (rel + 8 * (lane + 64 * k)) * BPS;
                const uint4 a = *reinterpret_cast<const uint4 *>(src);
                nxt[k].w[0] = a.x; nxt[k].w[1] = a.y; nxt[k].w[2] = a.z; nxt[k].w[3] = a.w;
                if (BPS == 4) {
                    const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
                    nxt[k].w[4 % RawGroup<FMT>::WORDS] = b.x; nxt[k].w[5 % RawGroup<FMT>::WORDS] = b.y;
                    nxt[k].w[6 % RawGroup<FMT>::WORDS] = b.z; nxt[k].w[7 % RawGroup<FMT>::WORDS] = b.w;
                }
                nxt_valid[k] = 0xffu;
            }
        } else {
#pragma unroll
            for (int k = 0; k < GPT; ++k)
                nxt_valid[k] = fetch_group<FMT>(P, (int64_t)P.batch_first + rel + 8 * (lane + 64 * k), nxt[k]);
        }
    };

    /* On the way: this wavefront's share of the previous batch's message records (its resolve and power kernels
     * ran before this launch), written inside one of its tiles from the candidate scratch -- a different tile for
     * neighbouring wavefronts, so that the PCIe writes of the 2 MB spread over the whole launch instead of
     * queueing up at its start (where every wavefront's next load would wait behind its own stores).  (Round 4: confining
     * the slices to the first 25-66 % of the tiles, so that the host-memory stores drain before the kernel ends, bought
     * nothing; 25 % was 15 % slower.) */
    const uint32_t emit_at = EMIT && P.emit.nbuffers && region / P.emit.stride < P.emit.nbuffers ? tile_lo + region % (tile_hi - tile_lo) : 0xffffffffu;

    TDECL
    for (uint32_t tile = tile_lo; tile < tile_hi; ++tile) {
        const uint64_t tile_pos0 = (uint64_t)tile * WT; /* first scan position, batch-relative */
        TMARK(8)
        const uint64_t a0 = P.batch_first + tile_pos0;

        /* ---- stage 1: IQ -> magnitudes in LDS; prefetch the next tile's IQ ---- */
        {
            const uint64_t c = tile_pos0 / MSD_CHUNK_SAMPLES;
            if (c != sum_chunk || sum_tiles == (P.tile_sums ? 1u : 16u / NH)) { /* wave-uniform */
                flush_sums();
                if (c != sum_chunk)
                    flush_chunk();
                sum_chunk = c;
                sum_tile0 = tile;
            }
            ++sum_tiles;
        }
        /* all of the tile's table loads first (8 per group, GPT groups), then their uses: the compiler keeps the
         * order it is given, and one round trip to the table instead of GPT is 3 us per tile */
        uint32_t mgs[GPT][8], mg_valid[GPT];
        if (MSD_PRIO_GATHER != MSD_PRIO_CONV)
            __builtin_amdgcn_s_setprio(MSD_PRIO_GATHER);
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            convert_group<FMT, false, SCAN_LUT>(cur[k], cur_valid[k], lut, mgs[k]);
            mg_valid[k] = cur_valid[k];
        }
        if (MSD_PRIO_GATHER != MSD_PRIO_CONV)
            __builtin_amdgcn_s_setprio(MSD_PRIO_CONV);
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            uint32_t(&mg)[8] = mgs[k];
            mask_group(mg_valid[k], mg);
            const uint4 packed = pack8(mg);
            *reinterpret_cast<uint4 *>(mags + FRONT + 8 * (lane + 64 * k)) = packed;
            if (P.mag_out) /* wave-uniform: Mode A/C is on, its candidate kernel reads these instead of the IQ */
                arena_store16(packed, reinterpret_cast<uint4 *>(P.mag_out + tile_pos0 + 8u * (uint32_t)(lane + 64 * k)));
            const uint32_t pk[4] = {packed.x, packed.y, packed.z, packed.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { /* two magnitudes per instruction */
                sum_level = dot2u(pk[i], 0x00010001u, sum_level);
                pw_mod = dot2u(pk[i], pk[i], pw_mod);
                const uint32_t top = pk_shr5(pk[i]);
                pw_top = dot2u(top, top, pw_top);
            }
        }
        wave_lds_sync();
        TMARK(0)

        /* the record slice goes here, behind the tile's table loads and in front of the next tile's prefetch: tests
         * and candidate rounds need no global data, so nothing waits for the PCIe stores until the next tile's
         * conversion, 15 us on */
        if (EMIT && tile == emit_at && !(P.debug_flags & 128)) /* wave-uniform */
            msd_emit_slice<FMT>(P.emit, P.lut, region, lane, X.w + W_HITS, (P.debug_flags & 64) != 0);

        /* ---- prefetch the next tile's IQ (behind the record slice: at that point neither this tile's raw groups nor
         * the next one's are live, which is what keeps the slice out of the loop's register budget) ---- */
        if (tile + 1 < tile_hi)
            fetch_tile(tile + 1);

        TMARK(1)
        if (!(P.debug_flags & 2)) {
            if (MSD_PRIO_TESTS != MSD_PRIO_CONV)
                __builtin_amdgcn_s_setprio(MSD_PRIO_TESTS);
            /* ---- stage 2: preamble tests for my NH runs of 16 consecutive positions (demod_2400.c:257-335) ---- */
            /* one bit plane per test and run, position q at bit 15 - q */
            uint32_t pl0[NH], pl1[NH], pl2[NH], any[NH], cnt[NH], rank0[NH];
            uint32_t H = 0;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                uint32_t v[20];
                {
                    const uint4 *src = reinterpret_cast<const uint4 *>(mags + 1024 * h + 16 * lane);
#if MSD_TESTS_SWZ
                    /* A lane's five 16-byte reads start 32 bytes after its neighbour's: in every group of sixteen lanes the
                     * LDS serves together (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...) the chunk numbers are all even or
                     * all odd, so half of the banks sit idle and the other half are asked twice or more.  The lanes 16-31
                     * and 48-63 of the wavefront therefore read their chunks in the order 1 0 3 2 4 while the others read
                     * 0 1 2 3 4 -- odd chunks beside even ones in every group -- and swap them back in registers. */
                    const bool odd_first = (lane & 16) != 0;
                    const int k0 = odd_first ? 1 : 0, k2 = odd_first ? 3 : 2;
                    const uint4 r0 = src[k0], r1 = src[k0 ^ 1], r2 = src[k2], r3 = src[k2 ^ 1], r4 = src[4];
                    const uint4 c0 = odd_first ? r1 : r0, c1 = odd_first ? r0 : r1, c2 = odd_first ? r3 : r2, c3 = odd_first ? r2 : r3;
                    v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
                    v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
                    v[8] = c2.x; v[9] = c2.y; v[10] = c2.z; v[11] = c2.w;
                    v[12] = c3.x; v[13] = c3.y; v[14] = c3.z; v[15] = c3.w;
                    v[16] = r4.x; v[17] = r4.y; v[18] = r4.z; v[19] = r4.w;
#else
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const uint4 q = src[k];
                        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
                    }
#endif
                }
#if MSD_TESTS_V2
                /* All 38 samples this run's 16 positions touch, unpacked once -- through opaque instructions, so that
                 * the compiler cannot fold the unpacking back into SDWA operands: on gfx950 a wave64 instruction with
                 * an SDWA / DPP / SGPR operand, a compare, a carry, a multiply or any three-operand integer form
                 * issues in 4 cycles, plain v_add / v_sub / v_and / v_or / v_lshrrev / v_ashrrev in 2
                 * (scripts/micro/valu_issue.hip) -- and every sample is an operand of a dozen of them. */
                int sm[40];
#pragma unroll
                for (int k = 0; k < 20; ++k) {
                    asm("v_and_b32 %0, 0xffff, %1" : "=v"(sm[2 * k]) : "v"(v[k]));
                    asm("v_lshrrev_b32 %0, 16, %1" : "=v"(sm[2 * k + 1]) : "v"(v[k]));
                }
                uint32_t p0 = 0, p1 = 0, p2 = 0;
#if MSD_TESTS_PRE_PLANE
                uint32_t ppre = 0; /* the pre-check's own plane: ANDed into the three others once per run, not once per position */
#endif
                const int thr = P.threshold, m32 = -32;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    /* pa[d] = mags[p + 2 + d].  Branch-free, and compare-free: every verdict is the sign bit of a
                     * difference (all values stay below 2^28), the pre-check the AND of three of them
                     * (demod_2400.c:276-282), one v_alignbit shifts a verdict into its plane.
                     *   ref - 1 = (base_noise * threshold - 32) >> 5 (arithmetic), common = sum_1_4 - diff_2_3 + pa9 + pa12,
                     *   test 0: common - diff_10_11 >= ref  <=>  (ref - 1) - common + diff_10_11 < 0, and so on. */
#define PA(d) (sm[q + 2 + (d)])
                    const int pre = (PA(7) - PA(1)) & (PA(14) - PA(12)) & (PA(15) - PA(12));
                    const int base_noise = PA(5) + PA(8) + PA(16) + PA(17) + PA(18);
                    int refm;
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(refm) : "v"(base_noise), "v"(thr), "v"(m32));
                    refm >>= 5;
                    const int diff_2_3 = PA(2) - PA(3), diff_10_11 = PA(10) - PA(11);
                    const int b = refm - (PA(1) + PA(4) + PA(12));
                    const int r1 = b + diff_2_3 - PA(9);
#if MSD_TESTS_PRE_PLANE
                    /* the sign of `pre & x` is the AND of the signs: four pushes and one three-way AND per position instead
                     * of three pushes, an AND and three three-way ANDs (7 issue cycles less per position) */
                    const int g0 = r1 + diff_10_11, g1 = r1 - diff_10_11;
                    const int g2 = b - diff_2_3 - diff_2_3 - diff_10_11;
                    ppre = __builtin_amdgcn_alignbit(ppre, (uint32_t)pre, 31);
#else
                    const int g0 = pre & (r1 + diff_10_11), g1 = pre & (r1 - diff_10_11);
                    const int g2 = pre & (b - diff_2_3 - diff_2_3 - diff_10_11);
#endif
#undef PA
                    p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)g0, 31); /* plane = 2 * plane + verdict */
                    p1 = __builtin_amdgcn_alignbit(p1, (uint32_t)g1, 31);
                    p2 = __builtin_amdgcn_alignbit(p2, (uint32_t)g2, 31);
                }
#if MSD_TESTS_PRE_PLANE
                p0 &= ppre;
                p1 &= ppre;
                p2 &= ppre;
#endif
#else
                /* all 36 samples this run's 16 positions touch, unpacked once */
                int sm[40];
#pragma unroll
                for (int k = 0; k < 20; ++k) {
                    sm[2 * k] = (int)(v[k] & 0xffffu);
                    sm[2 * k + 1] = (int)(v[k] >> 16);
                }
                uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    /* pa[d] = mags[p + 2 + d].  Branch-free on purpose: with 64 lanes some lane almost
                     * always passes the pre-check, so a branch only adds exec-mask bookkeeping. */
#define PA(d) (sm[q + 2 + (d)])
                    const uint64_t pre = __ballot(PA(1) > PA(7)) & __ballot(PA(12) > PA(14)) & __ballot(PA(12) > PA(15));
                    const uint32_t base_noise = (uint32_t)(PA(5) + PA(8) + PA(16) + PA(17) + PA(18));
                    const int ref_level = (int)(__umul24(base_noise, (uint32_t)P.threshold) >> 5); /* < 2^24 each */
                    const int diff_2_3 = PA(2) - PA(3);
                    const int sum_1_4 = PA(1) + PA(4);
                    const int diff_10_11 = PA(10) - PA(11);
                    const int common3456 = sum_1_4 - diff_2_3 + PA(9) + PA(12);
                    const uint64_t f0 = pre & __ballot(common3456 - diff_10_11 >= ref_level);
                    const uint64_t f1 = pre & __ballot(common3456 + diff_10_11 >= ref_level);
                    const uint64_t f2 = pre & __ballot(sum_1_4 + 2 * diff_2_3 + diff_10_11 + PA(12) >= ref_level);
#undef PA
                    MSD_PUSH(p0, f0);
                    MSD_PUSH(p1, f1);
                    MSD_PUSH(p2, f2);
                }
#endif
                /* positions past the last one the reference scans */
                {
                    const uint64_t first = a0 + 1024ull * h + 16ull * lane;
                    if (first + 16 > batch_end) {
                        const int keep = first >= batch_end ? 0 : (int)(batch_end - first);
                        const uint32_t km = keep ? ~((1u << (16 - keep)) - 1u) : 0u;
                        p0 &= km;
                        p1 &= km;
                        p2 &= km;
                    }
                }
                pl0[h] = p0;
                pl1[h] = p1;
                pl2[h] = p2;
                any[h] = p0 | p1 | p2;

                /* ---- stage 3: ranks of my hits among the wavefront's (position order: run, lane, bit) ---- */
                cnt[h] = (uint32_t)__popc(any[h]);
                const uint32_t incl = wave_incl_scan(cnt[h]);
                rank0[h] = H + incl - cnt[h];
                H += wave_last(incl);
            }

            TMARK(2)
            if (MSD_PRIO_CAND != MSD_PRIO_TESTS)
                __builtin_amdgcn_s_setprio(MSD_PRIO_CAND);
            if (!(P.debug_flags & 1) && H) {
                /* ---- stage 4: candidate rounds of up to HC hits ---- */
                uint32_t r0 = 0;
                bool fill = true;
                uint32_t nh = (H < (uint32_t)HC) ? H : (uint32_t)HC;
                while (r0 < H) {
                    if (fill) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            if (cnt[h] && rank0[h] < r0 + HC && rank0[h] + cnt[h] > r0) {
                                /* my hits with rank in [r0, r0 + HC) -> hitl, in position order */
                                uint32_t x = any[h], r = rank0[h];
                                while (x) {
                                    const int bit = 31 - __clz((int)x); /* highest bit = lowest position */
                                    x &= ~(1u << bit);
                                    if (r >= r0 && r < r0 + HC) {
                                        const uint32_t m = ((pl0[h] >> bit) & 1u) | (((pl1[h] >> bit) & 1u) << 1) |
                                                           (((pl2[h] >> bit) & 1u) << 2);
                                        hitl[r - r0] = (uint32_t)(1024 * h + 16 * lane + (15 - bit)) | (m << 13);
                                    }
                                    ++r;
                                }
                            }
                        }
                        wave_lds_sync();
                    }
                    const uint32_t out0 = hcur + r0;
                    const uint32_t room = out0 < P.hcap ? P.hcap - out0 : 0u;
                    TMARK(3)
                    if (!candidate_round<FIX2>(P, X, nh, tile_pos0, my_hits + out0, room, my_tries, tcur, try_base
#ifdef MSD_KERNEL_TIMING
                                               , tacc_, &tlast_
#endif
                                               )) {
                        nh = (nh + 1) / 2; /* too many tries with a known DF: halve the round */
                        fill = false;
                        wave_lds_sync();
                    } else {
                        r0 += nh;
                        nh = (H - r0 < (uint32_t)HC) ? (H - r0) : (uint32_t)HC;
                        fill = true;
                    }
                }
            }
            hcur += H;
            if (MSD_PRIO_CAND != MSD_PRIO_CONV)
                __builtin_amdgcn_s_setprio(MSD_PRIO_CONV);
        }
        TMARK(9)

        /* ---- carry the last 328 magnitudes over as the next tile's look-behind ---- */
        uint4 carry = make_uint4(0, 0, 0, 0);
        if (lane < FRONT / 8)
            carry = *reinterpret_cast<const uint4 *>(mags + WT + 8 * lane);
        wave_lds_sync();
        if (lane < FRONT / 8)
            *reinterpret_cast<uint4 *>(mags + 8 * lane) = carry;
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            cur[k] = nxt[k];
            cur_valid[k] = nxt_valid[k];
        }
    }
    TMARK(8)
    TFLUSH
    flush_sums();
    flush_chunk();
    hits_total = hcur;
    tries_total = tcur;
}

template <int FMT, bool FIX2 /* --aggressive: two-bit correction tables in global memory */,
          bool EMIT /* the wavefronts also write the previous batch's message records (P.emit): an instantiation of its
                       own, because the call alone costs the tile loop 7 us per launch in spills */>
#ifndef MSD_SCAN_OCC
#define MSD_SCAN_OCC ((MSD_SCAN_WAVES * MSD_SCAN_WGS_PER_CU + 3) / 4) /* wavefronts per SIMD the register budget is held to */
#endif
__global__ void __launch_bounds__(NT, MSD_SCAN_OCC) msd_scan_kernel(const MsdScanParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *syn = reinterpret_cast<uint32_t *>(smem + OFF_SYN);
    uint32_t *sl = reinterpret_cast<uint32_t *>(smem + OFF_SL);
    uint32_t *wgc = reinterpret_cast<uint32_t *>(smem + OFF_WGC);
    uint16_t *lds_lut = reinterpret_cast<uint16_t *>(smem + OFF_LUT);
    const uint16_t *lut = MSD_LUT_GLOBAL ? P.lut : lds_lut;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    /* constant tables -> LDS, once per workgroup */
    for (int i = tid; i < (int)MSD_SYNH_WORDS; i += NT)
        syn[i] = P.synhash[i];
    for (int i = tid; i < (int)MSD_SLICER_WORDS; i += NT)
        sl[i] = P.slicer[i];
    if (FMT == MSD_FMT_UC8 && !MSD_LUT_GLOBAL) {
        const uint4 *g = reinterpret_cast<const uint4 *>(P.lut);
        uint4 *l = reinterpret_cast<uint4 *>(lds_lut);
        for (int i = tid; i < 128 * LUT_STRIDE * 2 / 16; i += NT)
            l[i] = g[i];
    }
    __syncthreads();

    /* from here to the end of the batch the wavefronts do not meet again */
    const uint32_t region = blockIdx.x * WAVES + (uint32_t)wave;
    uint32_t tile_lo, tile_hi;
    if (P.regions_per_buffer) { /* lean layout: piece region % k of buffer region / k, never across a buffer boundary */
        constexpr uint32_t TPB = MSD_CHUNK_SAMPLES / (1024u * tile_runs(FMT)); /* tiles per buffer */
        const uint32_t b = region / P.regions_per_buffer, piece = region - b * P.regions_per_buffer;
        tile_lo = b * TPB + piece * P.tiles_per_region;
        tile_hi = tile_lo + P.tiles_per_region;
        if (tile_hi > (b + 1) * TPB)
            tile_hi = (b + 1) * TPB;
        if (tile_lo > tile_hi)
            tile_lo = tile_hi;
        if (region == 0 && P.tail_words) /* the batch's last samples, for the look-behind of its successor */
            for (uint32_t i = (uint32_t)lane; i < P.tail_words; i += 64)
                P.tail_dst[i] = P.tail_src[i];
    } else {
        tile_lo = region * P.tiles_per_wg;
        tile_hi = tile_lo + P.tiles_per_wg;
    }
    if (tile_hi > P.ntiles)
        tile_hi = P.ntiles;
    uint32_t nhits = 0, ntries = 0;
    if (tile_lo < tile_hi) { /* wave-uniform */
        WaveCtx X;
        X.w = smem + OFF_WAVE + wave * W_BYTES;
        X.syn = syn;
        X.sl = sl;
        X.lane = lane;
        scan_region<FMT, FIX2, EMIT>(P, X, lut, region, tile_lo, tile_hi, nhits, ntries);
    } else if (EMIT && P.emit.nbuffers && region / P.emit.stride < P.emit.nbuffers && !(P.debug_flags & 128)) {
        /* a region without tiles (lean layout: the pieces of a short last buffer) still owes its share of the
         * previous batch's records */
        msd_emit_slice<FMT>(P.emit, P.lut, region, lane, smem + OFF_WAVE + wave * W_BYTES + W_HITS, (P.debug_flags & 64) != 0);
    }
    if (lane == 0) {
        wgc[4 * wave] = nhits;
        wgc[4 * wave + 1] = ntries;
        wgc[4 * wave + 2] = (nhits > P.hcap || ntries > P.tcap) ? 1u : 0u;
    }
    __syncthreads();
    if (tid < WAVES) {
        msd_region_counts c = {};
        uint32_t hb = 0, tb = 0, ovf = 0;
        for (int i = 0; i < WAVES; ++i) {
            if (i == tid) {
                c.hbase = hb;
                c.tbase = tb;
            }
            hb += wgc[4 * i];
            tb += wgc[4 * i + 1];
            ovf |= wgc[4 * i + 2];
        }
        c.nhits = wgc[4 * tid];
        c.ntries = wgc[4 * tid + 1];
        c.overflow = wgc[4 * tid + 2];
        P.counts[blockIdx.x * WAVES + tid] = c;
        if (tid == 0) {
            msd_wg_totals t = {hb, tb, ovf, 0};
            P.wg_totals[blockIdx.x] = t;
            if (ovf && P.overflow)
                atomicOr(P.overflow, 1ull);
        }
    }
}

/* dense[offset + i] = region[i] for every region (= wavefront of the scan kernel); the try index inside a
 * hit record is made dense too.  grid = one workgroup per region; a region's offset is the totals of the
 * scan workgroups in front of its own plus what the scan kernel left in its counts. */
__global__ void __launch_bounds__(256) msd_gather_kernel(const msd_region_counts *counts, const msd_wg_totals *wgt,
                                                         const msd_hit *hits, const msd_try *tries, uint32_t hcap,
                                                         uint32_t tcap, msd_hit *dense_hits, uint64_t dense_hcap,
                                                         msd_try *dense_tries, uint64_t dense_tcap, uint64_t *totals,
                                                         uint64_t *sums, uint32_t nbuffers, uint64_t *h_totals,
                                                         uint64_t *h_sums, uint4 *wipe, uint32_t wipe_n,
                                                         const uint32_t *tail_src, uint32_t *tail_dst,
                                                         uint32_t tail_words, uint32_t region_len, uint32_t *buf_first,
                                                         uint32_t try_abs /* lean layout: the hits hold arena-absolute try indices */)
{
    __shared__ unsigned long long ph[4], pt[4];
    __shared__ uint32_t povf[4];
    const uint32_t w = blockIdx.x, nreg = gridDim.x, tid = threadIdx.x;
    const uint32_t my_wg = w / WAVES, nswg = (nreg + WAVES - 1) / WAVES;
    unsigned long long h = 0, t = 0;
    uint32_t ovf = 0;
    const bool last = w + 1 == nreg;
    for (uint32_t i = tid; i < (last ? nswg : my_wg); i += 256) { /* the last workgroup also owes the totals */
        const msd_wg_totals c = wgt[i];
        if (i < my_wg) {
            h += c.nhits;
            t += c.ntries;
        }
        ovf |= c.overflow;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        h += __shfl_down(h, d, 64);
        t += __shfl_down(t, d, 64);
        ovf |= __shfl_down(ovf, d, 64);
    }
    if ((tid & 63) == 0) {
        ph[tid >> 6] = h;
        pt[tid >> 6] = t;
        povf[tid >> 6] = ovf;
    }
    __syncthreads();
    const msd_region_counts mine = counts[w];
    const uint64_t ho = ph[0] + ph[1] + ph[2] + ph[3] + mine.hbase, to = pt[0] + pt[1] + pt[2] + pt[3] + mine.tbase;
    const uint32_t nh = mine.nhits < hcap ? mine.nhits : hcap;
    const uint32_t nt = mine.ntries < tcap ? mine.ntries : tcap;
    if (last) {
        /* list totals and per-buffer level/power sums straight to pinned host memory (a copy on another
         * stream would queue behind the following scans); the device sums are zeroed for the slot's next batch */
        if (tid == 0) {
            const uint64_t ah = ho + mine.nhits, at = to + mine.ntries;
            const uint64_t o = (povf[0] | povf[1] | povf[2] | povf[3]) ? 1 : 0;
            totals[0] = ah;
            totals[1] = at;
            totals[2] = o;
            if (h_totals) {
                h_totals[0] = ah;
                h_totals[1] = at;
                h_totals[2] = o;
            }
        }
        if (h_sums)
            for (uint32_t i = tid; i < 2 * nbuffers; i += 256) {
                h_sums[i] = sums[i];
                sums[i] = 0;
            }
    }
    if (w == 0) /* the last samples of the batch, kept for the look-behind of the next one */
        for (uint32_t i = tid; i < tail_words; i += 256)
            tail_dst[i] = tail_src[i];
    /* all-ones into a scratch table of the slot's resolve stage (the prediction table), spread over the grid */
    for (uint32_t i = w * blockDim.x + tid; i < wipe_n; i += nreg * blockDim.x)
        wipe[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    const msd_hit *hs = hits + (size_t)w * hcap;
    if (buf_first && tid < 64) {
        /* the buffers that start inside this region (the last region also answers for everything behind it, the
         * end sentinel included): where their hits start in the dense list, so that the resolve workgroups need
         * not search for it -- a 64-ary lower bound over the region's own hits */
        const uint64_t p0 = (uint64_t)w * region_len, p1 = last ? ~0ull : p0 + region_len;
        for (uint64_t bb = (p0 + MSD_CHUNK_SAMPLES - 1) / MSD_CHUNK_SAMPLES; bb <= nbuffers && bb * MSD_CHUNK_SAMPLES < p1; ++bb) {
            const uint64_t want = bb * MSD_CHUNK_SAMPLES;
            uint32_t lo = 0, hi = nh;
            while (lo < hi) {
                const uint32_t step = (hi - lo + 63) / 64, p = lo + tid * step;
                const bool below = p < hi && MSD_HIT_POS(hs[p]) < want;
                const uint32_t k = (uint32_t)__popcll(__ballot(below)); /* the first k probes are below: monotone */
                if (k == 0) {
                    hi = lo;
                } else {
                    const uint32_t nhi = lo + k * step;
                    lo = lo + (k - 1) * step + 1;
                    if (nhi < hi)
                        hi = nhi;
                }
            }
            if (tid == 0)
                buf_first[bb] = (uint32_t)(ho + lo < 0xffffffffull ? ho + lo : 0xffffffffull);
        }
    }
    for (uint32_t i = tid; i < nh; i += blockDim.x) {
        msd_hit hr = hs[i];
        if (MSD_HIT_NLIVE(hr))
            hr = (hr & ((1ull << 34) - 1)) | ((MSD_HIT_TRY(hr) - (try_abs ? (uint64_t)w * tcap : 0ull) + to) << 34);
        if (ho + i < dense_hcap)
            dense_hits[ho + i] = hr;
    }
    const uint4 *ts = reinterpret_cast<const uint4 *>(tries + (size_t)w * tcap);
    uint4 *td = reinterpret_cast<uint4 *>(dense_tries);
    for (uint32_t i = tid; i < 2 * nt; i += blockDim.x)
        if (to + (i >> 1) < dense_tcap)
            td[2 * to + i] = ts[i];
}


/* exclusive offsets of the per-workgroup regions in the dense lists; single workgroup */
__global__ void __launch_bounds__(256) msd_offsets_kernel(const msd_wg_counts *counts, uint32_t nwg,
                                                          uint64_t *offsets /* [nwg][2] */,
                                                          uint64_t *totals /* [4] */, uint64_t *sums,
                                                          uint32_t nbuffers, uint64_t *h_totals, uint64_t *h_sums)
{
    __shared__ unsigned long long sh[256], st[256];
    __shared__ uint32_t ovf;
    const int tid = threadIdx.x;
    if (tid == 0)
        ovf = 0;
    __syncthreads();
    /* thread t owns workgroups [t*per, (t+1)*per) */
    const uint32_t per = (nwg + 255) / 256;
    unsigned long long h = 0, t = 0;
    for (uint32_t i = tid * per; i < (tid + 1) * per && i < nwg; ++i) {
        h += counts[i].nhits;
        t += counts[i].ntries;
        if (counts[i].overflow)
            atomicOr(&ovf, 1u);
    }
    sh[tid] = h;
    st[tid] = t;
    __syncthreads();
    if (tid == 0) {
        unsigned long long ah = 0, at = 0;
        for (int i = 0; i < 256; ++i) {
            const unsigned long long x = sh[i], y = st[i];
            sh[i] = ah;
            st[i] = at;
            ah += x;
            at += y;
        }
        totals[0] = ah;
        totals[1] = at;
        totals[2] = ovf;
        if (h_totals) { /* what the host waits for goes straight to pinned memory (see msd_publish_kernel) */
            h_totals[0] = ah;
            h_totals[1] = at;
            h_totals[2] = ovf;
        }
    }
    if (h_sums)
        for (uint32_t i = tid; i < 2 * nbuffers; i += 256) {
            h_sums[i] = sums[i];
            sums[i] = 0; /* ready for the slot's next batch */
        }
    __syncthreads();
    h = sh[tid];
    t = st[tid];
    for (uint32_t i = tid * per; i < (tid + 1) * per && i < nwg; ++i) {
        offsets[2 * i] = h;
        offsets[2 * i + 1] = t;
        h += counts[i].nhits;
        t += counts[i].ntries;
    }
}


/* one magnitude of the stream, by absolute sample index (used by the small follow-up kernels) */
template <int FMT>
__device__ __forceinline__ uint32_t stream_mag(const MsdScanParams &P, int64_t n, const uint16_t *lut_g)
{
    MsdSampleSource S;
    S.iq = P.iq;
    S.prev_tail = P.prev_tail;
    S.have_prev = P.have_prev;
    S.batch_first = P.batch_first;
    S.nsamples = P.nsamples;
    return msd_stream_mag<FMT>(S, n, lut_g);
}

/* One wavefront writes its share of job J (the previous batch's message records): wavefront w of the scan takes
 * slice w % stride of buffer w / stride, a run of consecutive records, between two of its tiles.  Up to 32 records at
 * a time are put together in `lds` (the wavefront's candidate scratch) and leave as consecutive dwords -- the
 * destination is host memory, where a lane-strided struct store costs a PCIe write per piece.  Without J.rec_off the
 * place of the buffer's records in the batch's array is added up from the per-buffer counts.  (Summing the signal
 * power here as well -- three messages' samples in flight per wavefront -- was measured: the code alone costs the
 * tile loop 56 more bytes of spills and the launch 11 us, in use 32 us; the resolve workgroups do it instead.) */
template <int FMT>
__device__ inline void msd_emit_slice(const MsdEmitJob &J, const uint16_t *lut_g, uint32_t w, int lane, unsigned char *lds,
                                      bool dbg_no_store)
{
    const uint32_t b = w / J.stride, slice = w % J.stride;
    /* one round of loads: flags, the buffer's counts and clocks, the counts in front of it */
    const uint64_t ovf = J.totals[2], ac_ovf = J.ac ? J.ac_totals[2] : 0;
    const uint32_t nm = J.nmsgs[b], na_all = J.ac ? J.nac[b] : 0u;
    uint32_t o;
    if (J.rec_off) {
        o = J.rec_off[b];
    } else {
        uint32_t mine = 0;
        for (uint32_t i = (uint32_t)lane; i < b; i += 64)
            mine += J.nmsgs[i] + (J.ac ? J.nac[i] : 0u);
        o = wave_last(wave_incl_scan(mine));
    }
    if (ovf || ac_ovf)
        return; /* arenas overflowed: the host rescans the batch */
    const uint64_t sample_ts = J.ts[2 * b], sys_ts = J.ts[2 * b + 1];
    const uint32_t base = b * MSD_CHUNK_SAMPLES;
    const msd_acc *acc = J.acc + (size_t)b * MSD_RB_MSG_CAP;
    constexpr uint32_t ROUND = 32;
    msd_wire *rec = reinterpret_cast<msd_wire *>(lds);
    static_assert(sizeof(msd_wire) % 8 == 0 && ROUND * sizeof(msd_wire) <= W_BYTES - W_HITS, "the records of a round fit the candidate scratch");
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
        for (uint32_t m0 = lo; m0 < hi; m0 += ROUND) {
            const uint32_t m = m0 + (uint32_t)lane, cnt = min(ROUND, hi - m0);
            const bool mine = (uint32_t)lane < cnt;
            msd_acc a = acc[mine ? m : m0];
            const unsigned long long psum = J.power[(size_t)b * MSD_RB_MSG_CAP + (mine ? m : m0)];
            if (mine) {
                unsigned long long side;
                rec[lane].mm = msd_emit_mode_s(a, J.tries, psum, sample_ts, sys_ts, base, side);
                if (o + m < J.cap)
                    J.side[o + m] = side;
            }
            flush(o + m0, cnt);
        }
    }
    if (J.ac) { /* the buffer's Mode A/C replies follow its Mode S messages (readsb.c:826-829) */
        const uint32_t na = na_all;
        const uint32_t *acc_ac = J.acc_ac + (size_t)b * MSD_RB_AC_CAP;
        const uint32_t run = (na + J.stride - 1) / J.stride, lo = min(na, slice * run), hi = min(na, lo + run);
        for (uint32_t m0 = lo; m0 < hi; m0 += ROUND) {
            const uint32_t m = m0 + (uint32_t)lane, cnt = min(ROUND, hi - m0);
            if ((uint32_t)lane < cnt) {
                rec[lane].mm = msd_emit_mode_ac(J.ac[acc_ac[m]], sample_ts, sys_ts);
                if (o + nm + m < J.cap)
                    J.side[o + nm + m] = 0;
            }
            flush(o + nm + m0, cnt);
        }
    }
}

/* Signal power of the accepted messages (demod_2400.c:386-399): sum of m[j+19+k]^2 over
 * msglen*12/5 samples.  Only accepted messages need it, so it runs after the resolve stage on
 * their positions: one wavefront per message. */
template <int FMT>
__global__ void __launch_bounds__(256) msd_power_kernel(const MsdScanParams P, const uint64_t *req /* pos << 16 | len */,
                                                        uint32_t nreq, unsigned long long *out)
{
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nreq)
        return;
    const uint64_t rq = req[i];
    const int len = (int)(rq & 0xffffu);
    const int64_t n0 = (int64_t)P.batch_first + (int64_t)(rq >> 16) - (int64_t)MSD_OVERLAP + 19;
    unsigned long long acc = 0;
    for (int k = lane; k < len; k += 64) {
        const uint32_t x = stream_mag<FMT>(P, n0 + k, P.lut);
        acc += (unsigned long long)(x * x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        acc += __shfl_down(acc, o);
    if (lane == 0)
        out[i] = acc;
}

/* The same for the GPU resolve stage, which does not know the number of messages on the host when
 * it queues the kernel: PB_WGS workgroups per buffer walk the accepted-message records the resolve
 * kernel left for that buffer, one wavefront per message; out[buffer][MSD_RB_MSG_CAP]. */
#ifndef MSD_PB_WGS
#define MSD_PB_WGS 4
#endif
constexpr uint32_t PB_WGS = MSD_PB_WGS;  /* workgroups per buffer: a buffer rarely holds more than 96 messages */

template <int FMT>
__global__ void __launch_bounds__(256) msd_power_buffers_kernel(const MsdScanParams P, const msd_acc *acc,
                                                                const msd_try *tries, const uint32_t *nmsgs,
                                                                const uint64_t *totals, unsigned long long *out,
                                                                const uint32_t *nac, uint32_t nbuffers, uint32_t *rec_off)
{
    if (totals[2])
        return;
    if (blockIdx.x == gridDim.x - 1 && rec_off) {
        /* where each buffer's records start in the batch's record array (its Mode S messages, then its Mode A/C
         * replies): an exclusive prefix over the buffers, for whoever writes the records */
        __shared__ uint32_t wsum[4];
        const uint32_t per = (nbuffers + 255u) / 256u, i0 = threadIdx.x * per;
        uint32_t mine = 0;
        for (uint32_t k = 0; k < per; ++k)
            if (i0 + k < nbuffers)
                mine += nmsgs[i0 + k] + (nac ? nac[i0 + k] : 0u);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d)
                incl += up;
        }
        if ((threadIdx.x & 63) == 63)
            wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = incl - mine;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
            off += wsum[w];
        for (uint32_t k = 0; k < per; ++k)
            if (i0 + k < nbuffers) {
                rec_off[i0 + k] = off;
                off += nmsgs[i0 + k] + (nac ? nac[i0 + k] : 0u);
            }
    }
    const uint32_t b = blockIdx.x / PB_WGS, nm = nmsgs[b];
    const int lane = threadIdx.x & 63;
    /* six messages per trip, so that their record, sample and table loads overlap: the kernel is a chain
     * of dependent loads (record -> IQ bytes -> magnitude table) and nothing else; a buffer's ~70 messages
     * are one trip for its PB_WGS x 4 wavefronts */
    constexpr uint32_t PER = 6, STEP = 4 * PB_WGS;
    for (uint32_t m0 = (blockIdx.x % PB_WGS) * 4 + (threadIdx.x >> 6); m0 < nm; m0 += PER * STEP) {
        msd_acc rec[PER];
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            const uint32_t m = m0 + u * STEP;
            rec[u] = acc[(size_t)b * MSD_RB_MSG_CAP + (m < nm ? m : m0)];
            if (m >= nm)
                rec[u].len = 0;
        }
        /* len is 134 or 268: five independent loads per lane instead of a data-dependent loop */
        uint32_t x[PER][5];
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            const int64_t n0 = (int64_t)P.batch_first + (int64_t)rec[u].pos - (int64_t)MSD_OVERLAP + 19;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const int k = lane + 64 * v;
                const uint32_t mg = stream_mag<FMT>(P, n0 + (k < (int)rec[u].len ? k : 0), P.lut); /* unconditional load */
                x[u][v] = k < (int)rec[u].len ? mg : 0u;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            /* the sum of at most 320 squares < 2^32 as two 32-bit sums (low and high halves of the squares),
             * each reduced over the wavefront with DPP adds: no 64-bit shuffles through the LDS */
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const uint32_t sq = x[u][v] * x[u][v];
                lo += sq & 0xffffu;
                hi += sq >> 16;
            }
            const uint32_t lo_all = wave_last(wave_incl_scan(lo)), hi_all = wave_last(wave_incl_scan(hi));
            const uint32_t m = m0 + u * STEP;
            if (lane == 0 && m < nm)
                out[(size_t)b * MSD_RB_MSG_CAP + m] = (unsigned long long)lo_all + ((unsigned long long)hi_all << 16);
        }
    }
}

/* IQ -> magnitude only, for the iq_convert_fn-shaped entry point (convert.h:33-38): writes the
 * u16 magnitudes and accumulates the integer level/power sums (UC8). */
template <int FMT>
__global__ void __launch_bounds__(256) msd_convert_kernel(const uint8_t *iq, uint32_t nsamples,
                                                          const uint16_t *lut_g, uint16_t *mag,
                                                          unsigned long long *sums)
{
    __shared__ __attribute__((aligned(16))) uint16_t lut[(FMT == MSD_FMT_UC8) ? 128 * LUT_STRIDE : 8];
    if (FMT == MSD_FMT_UC8) {
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *l = reinterpret_cast<uint4 *>(lut);
        for (int i = threadIdx.x; i < 128 * LUT_STRIDE * 2 / 16; i += blockDim.x)
            l[i] = g[i];
        __syncthreads();
    }
    MsdScanParams P = {};
    P.iq = iq;
    P.nsamples = nsamples;
    P.batch_first = 0;
    P.have_prev = 0;
    P.lut = lut_g;
    P.ragged = iq + (size_t)(nsamples & ~7u) * (RawGroup<FMT>::WORDS / 2); /* the staging buffer is padded */
    unsigned long long sl = 0, sp = 0;
    const uint32_t ngroups = (nsamples + 7) / 8;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        RawGroup<FMT> r;
        const uint32_t valid = fetch_group<FMT>(P, (int64_t)g * 8, r);
        uint32_t mg[8];
        convert_group<FMT>(r, valid, lut, mg);
        for (int k = 0; k < 8; ++k) {
            if (g * 8 + k < nsamples)
                mag[g * 8 + k] = (uint16_t)mg[k];
            sl += mg[k];
            sp += (unsigned long long)(mg[k] * mg[k]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_down(sl, o);
        sp += __shfl_down(sp, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], sl);
        atomicAdd(&sums[1], sp);
    }
}

/* mean_level / mean_power of the float converters are *sequential* float sums (convert.c:228-252):
 * s = fl(s + x) for the 131072 magnitudes (and squares) of a buffer in order, and float addition is not
 * associative.  Added one after the other that is a dependent chain of 131072 v_add_f32 (about 11 cycles
 * each from a lone wavefront: 0.6 ms per buffer).  But the chain has structure: while s stays inside one
 * binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-23) and fl(s + x) = s + (floor(x/u) + r) * u, where the
 * rounding r depends on x and, in a tie, on the parity of s/u only.  An element is therefore a function
 *     S -> S + c(S mod 2),   S = s/u an integer,  c(0), c(1) two integers,
 * such functions compose to functions of the same form, and composition is associative: a wavefront
 * composes 16 elements per lane, scans the 64 lane functions in order (shuffles) and knows S after every
 * lane -- exactly, as long as S stays below 2^24.  Where it does not (the sum enters the next binade, about
 * fifteen times per buffer) the lanes in front of that point are applied, that lane's 16 elements are added
 * with real float additions, and the rest of the block starts over with the new u; the same for the first
 * samples of a buffer, while s is still tiny or zero.  Bit-identical to the sequential sum.
 * One workgroup per buffer: wavefronts 2..7 turn samples into level and power values in LDS (the correctly
 * rounded square root makes that the bigger half of the work), chunk by chunk and one chunk ahead; wavefront 0
 * sums the levels, wavefront 1 the powers. */
constexpr int MSD_FMT_MAGSQ = 100; /* internal source "format" of the float sums: f32 magnitude squares */
constexpr int FM_THREADS = 512, FM_PRODUCERS = FM_THREADS - 128, FM_PER = 8, FM_CHUNK = FM_PRODUCERS * FM_PER;
constexpr int FS_PER = 16, FS_BLOCK = 64 * FS_PER; /* elements per lane and per scan */
static_assert(FM_CHUNK % FS_BLOCK == 0, "a chunk is a whole number of scan blocks");
constexpr uint32_t FS_SAT = 1u << 26; /* increments saturate here: anything >= 2^24 means "left the binade" */

/* What element x (a non-negative float <= 1, as bits) adds to S when s has exponent e >= -7: base, plus one
 * more if `tie` and the sum in front of the rounding is odd.  Branch-free: x / u = m * 2^-sh with
 * sh = e - exponent(x) >= -7; t = m << 7 and sh + 7 keep every shift inside 0..31, and an x below u / 2
 * (sh > 24) counts as zero.  base is clamped so that sixteen of them cannot wrap. */
__device__ __forceinline__ void fsum_element(uint32_t xb, int e, uint32_t &base, uint32_t &tie)
{
    const uint32_t ef = xb >> 23;
    const int sh = e - (ef ? (int)ef - 127 : -126);
    const uint32_t m = sh > 24 ? 0u : ((xb & 0x7fffffu) | (ef ? 0x800000u : 0u));
    const uint32_t k = (uint32_t)(min(sh, 24) + 7), t = m << 7, P = 1u << k;
    const uint32_t a = t >> k, rem2 = (t & (P - 1u)) << 1;
    tie = rem2 == P ? ((a & 1u) ? 2u : 1u) : 0u; /* 1: rounds up after an odd S, 2: after an even S */
    base = min(a + (rem2 > P ? 1u : 0u), FS_SAT);
}

/* s += x[0] + ... in order, x = the registers of lane `who` (uniform loop, plain v_add_f32) */
__device__ __forceinline__ float fsum_sequential(float s, const float (&x)[FS_PER], int who)
{
#pragma unroll
    for (int k = 0; k < FS_PER; ++k) {
        const float t = __shfl(x[k], who, 64);
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(t));
    }
    return s;
}

/* s after the FS_BLOCK elements the wavefront holds (lane L: elements 16 L .. 16 L + 15); all lanes return it */
__device__ __forceinline__ float fsum_block(float s, const float (&x)[FS_PER], int lane)
{
    bool nz = false;
#pragma unroll
    for (int k = 0; k < FS_PER; ++k)
        nz |= x[k] != 0.0f;
    int from = 0; /* lanes in front of `from` are done */
    while (from < 64) {
        const uint64_t todo = ~0ull << from;
        const uint64_t nzmask = __ballot(nz) & todo;
        if (!nzmask)
            break; /* adding zeros changes nothing */
        const uint32_t sb = __float_as_uint(s);
        const int e = (int)(sb >> 23) - 127;
        if (e < -7) { /* zero or tiny: skip the zeros, add one lane's elements the slow way */
            const int z = __ffsll((unsigned long long)nzmask) - 1;
            s = fsum_sequential(s, x, z);
            from = z + 1;
            continue;
        }
        /* this lane's sixteen elements as one function: c0 = increment of S if S is even in front of them,
         * c1 if it is odd.  What an element adds is worked out first (independent instructions); the
         * chain through the parity is three instructions per element and parity. */
        uint32_t base[FS_PER], tie[FS_PER];
#pragma unroll
        for (int k = 0; k < FS_PER; ++k)
            fsum_element(__float_as_uint(x[k]), e, base[k], tie[k]);
        uint32_t c0 = 0, c1 = 1; /* running S relative to an even / odd start (c1 carries the start's 1) */
#pragma unroll
        for (int k = 0; k < FS_PER; ++k) {
            c0 += base[k] + ((tie[k] >> (~c0 & 1u)) & 1u); /* tie 1 fires on odd, 2 on even */
            c1 += base[k] + ((tie[k] >> (~c1 & 1u)) & 1u);
        }
        c1 -= 1u;
        if (lane < from)
            c0 = c1 = 0; /* already applied: the identity */
        c0 = min(c0, FS_SAT);
        c1 = min(c1, FS_SAT);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { /* ordered composition: (lanes .. L-d) then (L-d+1 .. L) */
            const uint32_t p0 = __shfl_up(c0, d, 64), p1 = __shfl_up(c1, d, 64);
            if (lane >= d) {
                const uint32_t n0 = p0 + ((p0 & 1u) ? c1 : c0), n1 = p1 + (((1u + p1) & 1u) ? c1 : c0);
                c0 = min(n0, FS_SAT);
                c1 = min(n1, FS_SAT);
            }
        }
        const uint32_t S0 = (sb & 0x7fffffu) | 0x800000u;
        const uint32_t S = S0 + ((S0 & 1u) ? c1 : c0); /* after this lane's elements, if nothing left the binade */
        const uint64_t bad = __ballot(S >= (1u << 24)) & todo;
        const int stop = bad ? __ffsll((unsigned long long)bad) - 1 : 64;
        const uint32_t Sb = stop ? (uint32_t)__shfl((int)S, stop - 1, 64) : S0; /* lanes < from are identities */
        s = __uint_as_float((sb & 0xff800000u) | (Sb & 0x7fffffu));
        if (stop < 64)
            s = fsum_sequential(s, x, stop);
        from = stop + 1;
    }
    return s;
}

template <int FMT>
__global__ void __launch_bounds__(FM_THREADS) msd_float_means_kernel(const uint8_t *iq, uint64_t nsamples,
                                                                     uint64_t buffer_len, uint32_t nbuffers,
                                                                     float *out /* [nbuffers][2] */)
{
    __shared__ __attribute__((aligned(16))) float vals[2][2][FM_CHUNK]; /* [chunk parity][level, power][sample] */
    const uint32_t b = blockIdx.x;
    if (b >= nbuffers)
        return;
    const uint64_t first = (uint64_t)b * buffer_len;
    uint64_t n64 = nsamples > first ? nsamples - first : 0;
    if (n64 > buffer_len)
        n64 = buffer_len;
    const uint32_t n = (uint32_t)n64;
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + first;
    const int tid = threadIdx.x;
    const uint32_t nchunks = (n + FM_CHUNK - 1) / FM_CHUNK;
    float sum = 0.0f; /* thread 0: level, thread 64: power */
    for (uint32_t c = 0; c <= nchunks; ++c) {
        if (tid >= 128) { /* producers: chunk c */
            if (c < nchunks) {
                const uint32_t p = (uint32_t)(tid - 128);
#pragma unroll
                for (int k = 0; k < FM_PER; ++k) { /* consecutive lanes read consecutive samples */
                    const uint32_t i = (uint32_t)k * FM_PRODUCERS + p, g = c * FM_CHUNK + i;
                    float m = 0.0f, magsq = 0.0f;
                    if (FMT == MSD_FMT_MAGSQ) { /* --dcfilter: the clamped squares msd_dcfilter_kernel left */
                        if (g < n) {
                            magsq = __uint_as_float(src[g]);
                            m = __builtin_sqrtf(magsq);
                        }
                    } else if (g < n) {
                        const uint32_t w = src[g];
                        const int I = (int)(int16_t)(w & 0xffffu), Q = (int)(int16_t)(w >> 16);
                        const float fi = (float)I * inv, fq = (float)Q * inv;
                        const float sq_i = fi * fi, sq_q = fq * fq;
                        magsq = sq_i + sq_q;
                        magsq = fminf(magsq, 1.0f); /* convert.c: if (magsq > 1) magsq = 1 -- the sum of two squares is not a NaN */
                        m = msd_sqrt_cr(magsq);
                    }
                    vals[c & 1][0][i] = m;
                    vals[c & 1][1][i] = magsq;
                }
            }
        } else if (c > 0) { /* wavefront 0: levels, wavefront 1: powers; chunk c - 1 */
            const float *v = vals[(c - 1) & 1][tid >> 6];
            const int lane = tid & 63;
#pragma unroll 1
            for (int blk = 0; blk < FM_CHUNK / FS_BLOCK; ++blk) {
                float x[FS_PER];
                const float4 *v4 = reinterpret_cast<const float4 *>(v + blk * FS_BLOCK + lane * FS_PER);
#pragma unroll
                for (int k = 0; k < FS_PER / 4; ++k) {
                    const float4 q = v4[k];
                    x[4 * k] = q.x; x[4 * k + 1] = q.y; x[4 * k + 2] = q.z; x[4 * k + 3] = q.w;
                }
                sum = fsum_block(sum, x, lane);
            }
        }
        __syncthreads();
    }
    if ((tid & 63) == 0 && tid < 128)
        out[2 * b + (tid >> 6)] = sum;
}

/* ---- the same sums with the whole workgroup working on them ----
 * msd_float_means_kernel leaves a buffer's two sums to one wavefront each, 128 blocks of 1024 elements one
 * after the other.  A block's 1024 elements are ONE function of the same S -> S + c(S mod 2) form (the
 * ordered composition of its lanes' functions), valid while the sum stays in the binade the function was
 * built for -- and the binade at the start of a block can be predicted: the sums only grow, so an ordinary
 * (unordered, approximate) prefix sum of the blocks' totals tells the exponent of the exact sum at every
 * block boundary unless that sum sits within rounding of a power of two.  So:
 *   pass 1  all eight wavefronts: approximate total of every block (level and power)
 *   prefix  predicted exponent at the start of every block; blocks inside which the exponent changes
 *           (about one per binade the sum passes through) and the first ones (tiny sums) are "slow"
 *   pass 2  all eight wavefronts: the composite function (c0, c1) of every other block under its
 *           predicted exponent
 *   apply   one wavefront per sum walks the 128 blocks in order: a block whose prediction holds (exponent
 *           as predicted, result still inside the binade) is one add; slow blocks and mispredictions are
 *           summed exactly with fsum_block.  Bit-identical to the sequential sum by construction: every
 *           shortcut is verified against the exact state before it is used. */
constexpr int FB_MAX = 128; /* blocks of FS_BLOCK elements per buffer (MSD_CHUNK_SAMPLES / 1024) */
constexpr int FM_SLOTS = 24; /* slow blocks per sum whose sub-block totals are kept (about ten occur) */

/* s + x(lane 0) + x(lane 1) + ... + x(lane 63), one addition after the other (convert.c:241-242).  The 64 values go
 * through 256 bytes of the wavefront's LDS and come back to every lane, sixteen broadcast reads of four: the chain is
 * then 64 dependent additions with nothing between them.  (A v_readlane per element in front of its addition: 19 cycles
 * per element; the sum travelling from lane to lane by DPP wave_shr:1, 63 additions and no reads at all: slower still.) */
__device__ __forceinline__ float fsum_lanes_in_order(float s, float x, float *lds64, int lane)
{
    lds64[lane] = x;
    wave_lds_sync();
    float4 q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        q[i] = reinterpret_cast<const float4 *>(lds64)[i];
    wave_lds_sync(); /* the next call's stores come behind these loads */
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(q[i].x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(q[i].y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(q[i].z));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(q[i].w));
    }
    return s;
}

/* a sample word as the level sum (which = 0) or the power sum (1) sees it */
template <int FMT>
__device__ __forceinline__ float fm_word_value(uint32_t w, float inv, int which)
{
    float m, magsq;
    if (FMT == MSD_FMT_MAGSQ) { /* --dcfilter: the clamped squares msd_dcfilter_kernel left */
        magsq = __uint_as_float(w);
        m = __builtin_sqrtf(magsq);
    } else {
        const float fi = (float)(int)(int16_t)(w & 0xffffu) * inv, fq = (float)(int)(int16_t)(w >> 16) * inv;
        const float sq_i = fi * fi, sq_q = fq * fq;
        magsq = sq_i + sq_q;
        magsq = fminf(magsq, 1.0f); /* convert.c: if (magsq > 1) magsq = 1 -- the sum of two squares is not a NaN */
        m = msd_sqrt_cr(magsq);
    }
    return which ? magsq : m;
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

/* ---- the same again as three launches (round 4) ----
 * In msd_float_means2_kernel a buffer belongs to one workgroup from beginning to end: pass 2 (vector-ALU work, the
 * correctly rounded square root of every sample) runs at four wavefronts per SIMD with every load's latency exposed, and
 * while two wavefronts walk the blocks in order the other six wait.  Here the parts are kernels of their own:
 *   msd_fm_totals_kernel     only without a scan in front (--dcfilter, the converter entry): approximate block totals
 *   msd_fm_functions_kernel  pass 2 for all blocks of all buffers, PARTS workgroups per buffer, at most 64 registers
 *                            (eight wavefronts per SIMD): the samples are converted on the way through -- no arrays of
 *                            values -- and converted again in the few blocks that need the general function
 *   msd_fm_apply_kernel      two wavefronts per buffer: the first block from zero, then the walk
 * with the predictions, functions and sub-block totals in an FmBufWork per buffer in device memory. */
struct FmBufWork {
    int e[2][FB_MAX], ca[2][FB_MAX];          /* predicted exponent at the block's start / first exponent of its sub-block totals */
    uint32_t f0[2][FB_MAX], f1[2][FB_MAX];    /* the block's function under e */
    uint32_t slot[2][FB_MAX];                 /* where its sub-block totals are; 0xff: none */
    uint32_t sub[2][FM_SLOTS][2][2][16];      /* [sum][slot][exponent ca, ca + 1][S even, odd][sub-block]: its function */
    float tot[2 * FB_MAX];                    /* approximate totals (level, power interleaved) when no scan left them */
};
constexpr int FMK_THREADS = 256, FMK_WAVES = FMK_THREADS / 64;
#ifndef MSD_FM_OCC
#define MSD_FM_OCC 8 /* wavefronts per SIMD the functions kernel is held to */
#endif
#ifndef MSD_FM_PARTS
#define MSD_FM_PARTS 4
#endif
constexpr int FM_PARTS = MSD_FM_PARTS, FM_BPP = FB_MAX / FM_PARTS; /* workgroups per buffer, blocks per workgroup */

/* one sample word as (level value, power value): convert.c:228-240 (SC16), :345-357 (SC16Q11); zero words give zeros */
template <int FMT>
__device__ __forceinline__ void fm_convert(uint32_t w, float inv, float &m, float &magsq)
{
    if (FMT == MSD_FMT_MAGSQ) { /* --dcfilter: the clamped squares msd_dcfilter_kernel left */
        magsq = __uint_as_float(w);
        m = __builtin_sqrtf(magsq);
    } else {
        const float fi = (float)(int)(int16_t)(w & 0xffffu) * inv, fq = (float)(int)(int16_t)(w >> 16) * inv;
        const float sq_i = fi * fi, sq_q = fq * fq;
        magsq = sq_i + sq_q;
        magsq = fminf(magsq, 1.0f); /* convert.c: if (magsq > 1) magsq = 1 -- the sum of two squares is not a NaN */
        m = msd_sqrt_cr(magsq);
    }
}

/* the block's sixteen words of this lane (elements 16 L .. 16 L + 15), zeros past the buffer's end */
__device__ __forceinline__ void fm_block_words(const uint32_t *src, uint32_t n, uint32_t blk, int lane, uint32_t (&w)[FS_PER])
{
    const uint32_t g0 = blk * FS_BLOCK + (uint32_t)lane * FS_PER;
    if (g0 + FS_PER <= n) {
        const uint4 *q = reinterpret_cast<const uint4 *>(src + g0);
#pragma unroll
        for (int k = 0; k < FS_PER / 4; ++k) {
            const uint4 v = q[k];
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < FS_PER; ++k)
            w[k] = g0 + k < n ? src[g0 + k] : 0u;
    }
}

template <int FMT>
__global__ void __launch_bounds__(FMK_THREADS) msd_fm_totals_kernel(const uint8_t *iq, uint64_t nsamples, uint64_t buffer_len,
                                                                    uint32_t nbuffers, FmBufWork *work)
{
    const uint32_t b = blockIdx.x / FM_PARTS, part = blockIdx.x % FM_PARTS;
    const uint64_t first = (uint64_t)b * buffer_len;
    uint64_t n64 = nsamples > first ? nsamples - first : 0;
    if (n64 > buffer_len)
        n64 = buffer_len;
    const uint32_t n = (uint32_t)n64;
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + first;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nblk = (n + FS_BLOCK - 1) / FS_BLOCK;
    const uint32_t hi = min(nblk, (part + 1u) * FM_BPP);
    for (uint32_t blk = part * FM_BPP + (uint32_t)wave; blk < hi; blk += FMK_WAVES) {
        uint32_t w[FS_PER];
        fm_block_words(src, n, blk, lane, w);
        float sl = 0.0f, sp = 0.0f;
#pragma unroll
        for (int k = 0; k < FS_PER; ++k) {
            float magsq;
            if (FMT == MSD_FMT_MAGSQ) {
                magsq = __uint_as_float(w[k]);
            } else { /* I^2 + Q^2 as one v_dot2 on the packed sample: a relative 2^-23 off the float path, a prediction does not care */
                typedef short short2_t __attribute__((ext_vector_type(2)));
                const short2_t v = __builtin_bit_cast(short2_t, w[k]);
                const uint32_t d = (uint32_t)__builtin_amdgcn_sdot2(v, v, 0, false); /* <= 2^31 */
                magsq = fminf((float)d * (inv * inv), 1.0f);
            }
            sl += __builtin_amdgcn_sqrtf(magsq);
            sp += magsq;
        }
        sl = wave_sum_f32(sl);
        sp = wave_sum_f32(sp);
        if (lane == 0) {
            work[b].tot[2 * blk] = sl;
            work[b].tot[2 * blk + 1] = sp;
        }
    }
}

template <int FMT>
__global__ void __launch_bounds__(FMK_THREADS, MSD_FM_OCC) msd_fm_functions_kernel(const uint8_t *iq, uint64_t nsamples, uint64_t buffer_len,
                                                                          uint32_t nbuffers, FmBufWork *work,
                                                                          const float *tile_sums /* or NULL: work[].tot */)
{
    __shared__ float blk_tot[2][FB_MAX];
    __shared__ int blk_e[2][FB_MAX], blk_ca[2][FB_MAX];
    __shared__ uint32_t blk_slot[2][FB_MAX];
    __shared__ uint32_t fm_next_blk;
    const uint32_t b = blockIdx.x / FM_PARTS, part = blockIdx.x % FM_PARTS;
    const uint64_t first = (uint64_t)b * buffer_len;
    uint64_t n64 = nsamples > first ? nsamples - first : 0;
    if (n64 > buffer_len)
        n64 = buffer_len;
    const uint32_t n = (uint32_t)n64;
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + first;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nblk = (n + FS_BLOCK - 1) / FS_BLOCK; /* <= FB_MAX */
    constexpr int SLOW = -2147483647 - 1;
    FmBufWork &W = work[b];
    {
        const float *ts = tile_sums ? tile_sums + 2 * (first / FS_BLOCK) : W.tot;
        for (uint32_t blk = (uint32_t)tid; blk < nblk; blk += FMK_THREADS) {
            blk_tot[0][blk] = ts[2 * blk];
            blk_tot[1][blk] = ts[2 * blk + 1];
        }
        if (tid == 0)
            fm_next_blk = part ? part * FM_BPP : 1u; /* the buffer's first block is the apply kernel's */
    }
    __syncthreads();
    /* predictions (every workgroup of the buffer works them out; part 0 leaves them for the apply kernel): wavefront 0
     * the levels, wavefront 1 the powers, lane L has blocks 2 L and 2 L + 1 -- see msd_float_means2_kernel */
    if (wave < 2) {
        const uint32_t b0 = 2u * (uint32_t)lane, b1 = b0 + 1u;
        const float t0 = b0 < nblk ? blk_tot[wave][b0] : 0.0f, t1 = b1 < nblk ? blk_tot[wave][b1] : 0.0f;
        float incl = t0 + t1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float up = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += up;
        }
        const float run0 = incl - (t0 + t1) < 0.0f ? 0.0f : incl - (t0 + t1);
        const float runs[2] = {run0, run0 + t0}, tots[2] = {t0, t1};
        int cas[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float run = runs[j], next = run + tots[j];
            const int e0 = (int)(__float_as_uint(run) >> 23) - 127, e1 = (int)(__float_as_uint(next) >> 23) - 127;
            const uint32_t mant = __float_as_uint(run) & 0x7fffffu;
            const bool near_edge = mant < 0x2000u || mant > 0x7fe000u;
            const bool slow = e0 != e1 || e0 < -7 || near_edge || tots[j] == 0.0f;
            int ca = e0 != e1 ? e0 : (near_edge && mant < 0x2000u ? e0 - 1 : e0);
            if (!slow || ca < -7 || tots[j] == 0.0f || b0 + j >= nblk)
                ca = SLOW;
            cas[j] = ca;
            if (b0 + j < nblk) {
                blk_e[wave][b0 + j] = slow ? SLOW : e0;
                if (part == 0)
                    W.e[wave][b0 + j] = slow ? SLOW : e0;
            }
        }
        const uint32_t mine_n = (cas[0] != SLOW ? 1u : 0u) + (cas[1] != SLOW ? 1u : 0u);
        uint32_t at = wave_incl_scan(mine_n) - mine_n;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (b0 + j < nblk) {
                const bool has = cas[j] != SLOW && at < (uint32_t)FM_SLOTS;
                blk_ca[wave][b0 + j] = has ? cas[j] : SLOW;
                blk_slot[wave][b0 + j] = has ? at : 0xffu;
                if (part == 0) {
                    W.ca[wave][b0 + j] = has ? cas[j] : SLOW;
                    W.slot[wave][b0 + j] = has ? at : 0xffu;
                }
                at += cas[j] != SLOW ? 1u : 0u;
            }
    }
    __syncthreads();
    const uint32_t blk_hi = min(nblk, (part + 1u) * FM_BPP);
    for (;;) {
        uint32_t blk = 0;
        if (lane == 0)
            blk = atomicAdd(&fm_next_blk, 1u);
        blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)blk);
        if (blk >= blk_hi)
            break;
        const int es[2] = {blk_e[0][blk], blk_e[1][blk]};
        const int cs[2] = {blk_ca[0][blk], blk_ca[1][blk]};
        if (es[0] == SLOW && es[1] == SLOW && cs[0] == SLOW && cs[1] == SLOW)
            continue; /* wave-uniform */
        uint32_t w[FS_PER];
        fm_block_words(src, n, blk, lane, w);
        /* The lane's sixteen elements as the function S -> S + c(S mod 2), with the adder itself: the sequential sum from
         * 2^e (an even S) and from 2^e + u (an odd one) -- while a sum stays inside the binade every addition rounds to a
         * multiple of u exactly as the real one does, ties to even included, and what is left above 2^e is the increment.  A
         * chain that leaves the binade stays outside (the elements are not negative) and then holds at least 2^23 units:
         * "leaves the binade" for the apply kernel, whatever the sum in front of the block.  Two dependent chains of
         * sixteen 2-cycle additions per sum, instead of an integer rounding per element, a tie test and a second
         * conversion of the block where a tie was found (the power sums of a quiet band tie in every block).  A sum
         * without a prediction runs under a stand-in exponent; its function is not used. */
        float A[2], acc0[2], acc1[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int e = es[s] != SLOW ? es[s] : 1; /* >= -7 */
            A[s] = __uint_as_float((uint32_t)(e + 127) << 23);
            acc0[s] = A[s];
            acc1[s] = A[s] + __uint_as_float((uint32_t)(e - 23 + 127) << 23);
        }
#pragma unroll
        for (int k = 0; k < FS_PER; ++k) {
            float m, magsq;
            fm_convert<FMT>(w[k], inv, m, magsq);
            acc0[0] += m;
            acc1[0] += m;
            acc0[1] += magsq;
            acc1[1] += magsq;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int e = es[s];
            if (e != SLOW) { /* wave-uniform */
                const float scale = __uint_as_float((uint32_t)(23 - e + 127) << 23), sat = (float)FS_SAT;
                uint32_t c0 = (uint32_t)fminf((acc0[s] - A[s]) * scale, sat);
                uint32_t c1 = (uint32_t)fminf((acc1[s] - A[s]) * scale, sat) - 1u; /* the start's own unit is not part of the increment */
                uint32_t f0, f1;
                if (!__ballot(c0 != c1)) { /* no lane's elements care about the parity: the block adds its total */
                    f0 = f1 = wave_last(wave_incl_scan(min(c0, 1u << 25)));
                } else {
                    /* Only the composition of all 64 lanes is wanted: a tree -- lanes 2d-1 mod 2d take in the lane d below them
                     * (DPP row_shr 1, 2, 4, 8), then lane 31 the row below and lane 63 lane 31 (row_bcast 15 / 31) -- with
                     * (earlier lanes) first, (this lane's) second in every step.  Six DPP pairs and no LDS round trip, where
                     * the scan through __shfl_up was twelve ds_bpermute, each waited for; the other lanes compute along and
                     * are not looked at (a missing source reads 0 = the identity). */
#define FM_COMPOSE(CTRL, RMASK)                                                                                         \
    {                                                                                                                   \
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c0, CTRL, RMASK, 0xf, false);                 \
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c1, CTRL, RMASK, 0xf, false);                 \
        const uint32_t n0 = p0 + ((p0 & 1u) ? c1 : c0), n1 = p1 + (((1u + p1) & 1u) ? c1 : c0);                        \
        c0 = min(n0, FS_SAT);                                                                                           \
        c1 = min(n1, FS_SAT);                                                                                           \
    }
                    FM_COMPOSE(0x111, 0xf) /* row_shr:1 */
                    FM_COMPOSE(0x112, 0xf) /* row_shr:2 */
                    FM_COMPOSE(0x114, 0xf) /* row_shr:4 */
                    FM_COMPOSE(0x118, 0xf) /* row_shr:8 */
                    FM_COMPOSE(0x142, 0xa) /* row_bcast:15 into rows 1 and 3 */
                    FM_COMPOSE(0x143, 0xc) /* row_bcast:31 into rows 2 and 3 */
#undef FM_COMPOSE
                    f0 = (uint32_t)__builtin_amdgcn_readlane((int)c0, 63);
                    f1 = (uint32_t)__builtin_amdgcn_readlane((int)c1, 63);
                }
                if (lane == 0) {
                    W.f0[s][blk] = f0;
                    W.f1[s][blk] = f1;
                }
            }
            const int ca = cs[s];
            if (ca != SLOW) { /* wave-uniform: the functions of the sixteen 64-sample sub-blocks (lanes 4 j .. 4 j + 3) for the
                                 exponents ca and ca + 1 -- the same chains, composed over four lanes */
                const uint32_t slot = blk_slot[s][blk];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int e = ca + c;
                    const float A2 = __uint_as_float((uint32_t)(e + 127) << 23);
                    float a0 = A2, a1 = A2 + __uint_as_float((uint32_t)(e - 23 + 127) << 23);
#pragma unroll
                    for (int k = 0; k < FS_PER; ++k) {
                        float m, magsq;
                        fm_convert<FMT>(w[k], inv, m, magsq);
                        a0 += s ? magsq : m;
                        a1 += s ? magsq : m;
                    }
                    const float scale = __uint_as_float((uint32_t)(23 - e + 127) << 23), sat = (float)FS_SAT;
                    uint32_t c0 = (uint32_t)fminf((a0 - A2) * scale, sat);
                    uint32_t c1 = (uint32_t)fminf((a1 - A2) * scale, sat) - 1u;
#pragma unroll
                    for (int d = 1; d < 4; d <<= 1) {
                        const uint32_t p0 = __shfl_up(c0, d, 64), p1 = __shfl_up(c1, d, 64);
                        if ((lane & 3) >= d) {
                            const uint32_t n0 = p0 + ((p0 & 1u) ? c1 : c0), n1 = p1 + (((1u + p1) & 1u) ? c1 : c0);
                            c0 = min(n0, FS_SAT);
                            c1 = min(n1, FS_SAT);
                        }
                    }
                    if ((lane & 3) == 3) {
                        W.sub[s][slot][c][0][lane >> 2] = c0;
                        W.sub[s][slot][c][1][lane >> 2] = c1;
                    }
                }
            }
        }
    }
}

template <int FMT>
__global__ void __launch_bounds__(128) msd_fm_apply_kernel(const uint8_t *iq, uint64_t nsamples, uint64_t buffer_len,
                                                           uint32_t nbuffers, const FmBufWork *work, float *out /* [nbuffers][2] */)
{
    __shared__ uint32_t slow_sub[2][FM_SLOTS][64]; /* [exponent][parity][sub-block] */
    __shared__ __attribute__((aligned(16))) float seq_vals[2][64];
    const uint32_t b = blockIdx.x;
    const uint64_t first = (uint64_t)b * buffer_len;
    uint64_t n64 = nsamples > first ? nsamples - first : 0;
    if (n64 > buffer_len)
        n64 = buffer_len;
    const uint32_t n = (uint32_t)n64;
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + first;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nblk = (n + FS_BLOCK - 1) / FS_BLOCK;
    constexpr int SLOW = -2147483647 - 1;
    const FmBufWork &W = work[b];
    if (n == 0) { /* a batch that ends on a buffer boundary counts one more, empty buffer: its samples are not there to read */
        if (lane == 0)
            out[2 * b + wave] = 0.0f;
        return;
    }
    uint32_t w[16];
    auto fetch = [&](uint32_t blk) { /* lane L: sample 64 j + L of the block in w[j], one per sub-block */
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t g = blk * (uint32_t)FS_BLOCK + 64u * j + (uint32_t)lane;
            w[j] = src[g < n ? g : 0u]; /* unconditional: sixteen loads in flight */
        }
    };
    fetch(0u); /* the first block's samples are on their way while the tables are read */
    /* the blocks' predictions and functions in registers, lane L: blocks L and L + 64; the sub-block totals in LDS */
    int re[2], rca[2];
    uint32_t rf0[2], rf1[2], rslot[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t blk = (uint32_t)lane + 64u * j;
        const bool in = blk < nblk;
        re[j] = in ? W.e[wave][blk] : SLOW;
        rca[j] = in ? W.ca[wave][blk] : SLOW;
        rslot[j] = in ? W.slot[wave][blk] : 0xffu;
        rf0[j] = in && re[j] != SLOW ? W.f0[wave][blk] : 0u; /* a slow block's function was never written */
        rf1[j] = in && re[j] != SLOW ? W.f1[wave][blk] : 0u;
    }
    {
        const uint32_t used = min((uint32_t)FM_SLOTS, (uint32_t)__popcll(__ballot(rca[0] != SLOW)) + (uint32_t)__popcll(__ballot(rca[1] != SLOW)));
        const uint32_t *gs = &W.sub[wave][0][0][0][0];
        uint32_t *ls = &slow_sub[wave][0][0];
        for (uint32_t i = (uint32_t)lane; i < used * 64u; i += 64u)
            ls[i] = gs[i];
    }
    const unsigned long long walk_lo = __ballot(re[0] == SLOW && rca[0] != SLOW), walk_hi = __ballot(re[1] == SLOW && rca[1] != SLOW);
    auto next_walk = [&](uint32_t from) -> uint32_t { /* first block with sub-block totals at or behind `from`; nblk: none */
        if (from < 64u) {
            const unsigned long long m = walk_lo >> from;
            if (m)
                return from + (uint32_t)__builtin_ctzll(m);
            from = 64u;
        }
        if (from < 128u) {
            const unsigned long long m = walk_hi >> (from - 64u);
            if (m)
                return from + (uint32_t)__builtin_ctzll(m);
        }
        return nblk;
    };
    uint32_t w_blk = next_walk(1u);
    /* The composite functions of the runs of blocks between two slow ones, a segmented scan over the lanes: the walk
     * then takes such a run in one step.  A head is a block that starts a run: every slow block (an identity on its own),
     * a block behind a slow one or behind one with another exponent, block 64 (the two registers are scanned apart). */
    uint32_t sf0[2], sf1[2];
    unsigned long long heads[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int up = __shfl_up(re[j], 1, 64);
        const int e_prev = lane ? up : SLOW;
        const bool slow = re[j] == SLOW || (j == 0 && lane == 0); /* block 0 is summed apart */
        bool flag = slow || e_prev != re[j] || (j == 0 && lane == 1);
        uint32_t c0 = slow ? 0u : min(rf0[j], FS_SAT), c1 = slow ? 0u : min(rf1[j], FS_SAT);
        heads[j] = __ballot(flag);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t p0 = __shfl_up(c0, d, 64), p1 = __shfl_up(c1, d, 64);
            const bool pflag = __shfl_up((int)flag, d, 64) != 0;
            if (lane >= d && !flag) {
                const uint32_t n0 = p0 + ((p0 & 1u) ? c1 : c0), n1 = p1 + (((1u + p1) & 1u) ? c1 : c0);
                c0 = min(n0, FS_SAT);
                c1 = min(n1, FS_SAT);
                flag = pflag;
            }
        }
        sf0[j] = c0;
        sf1[j] = c1;
    }
    auto next_head = [&](uint32_t from) -> uint32_t { /* first head at or behind `from`; 128: none */
        if (from < 64u) {
            const unsigned long long m = heads[0] >> from;
            if (m)
                return from + (uint32_t)__builtin_ctzll(m);
            from = 64u;
        }
        if (from < 128u) {
            const unsigned long long m = heads[1] >> (from - 64u);
            if (m)
                return from + (uint32_t)__builtin_ctzll(m);
        }
        return 128u;
    };
    uint32_t sb;
    { /* the buffer's first block starts from a sum of exactly zero and passes through a dozen binades: 1024 additions in order */
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t g = 64u * j + (uint32_t)lane;
            const float x = g < n ? fm_word_value<FMT>(w[j], inv, wave) : 0.0f;
            sum = fsum_lanes_in_order(sum, x, seq_vals[wave], lane);
        }
        sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum));
    }
    if (w_blk < nblk)
        fetch(w_blk);
    wave_lds_sync();
    for (uint32_t blk = 1; blk < nblk; ++blk) {
        const int hi = blk >= 64u, l = (int)(blk & 63u);
        const int e = __builtin_amdgcn_readlane(hi ? re[1] : re[0], l);
        bool fast = e != SLOW && (int)(sb >> 23) - 127 == e;
        if (fast && (((hi ? heads[1] : heads[0]) >> l) & 1ull)) { /* the whole run of blocks this one starts */
            const uint32_t last = min(next_head(blk + 1u), nblk) - 1u;
            const int lh = last >= 64u, ll = (int)(last & 63u);
            const uint32_t rf0s = (uint32_t)__builtin_amdgcn_readlane((int)(lh ? sf0[1] : sf0[0]), ll);
            const uint32_t rf1s = (uint32_t)__builtin_amdgcn_readlane((int)(lh ? sf1[1] : sf1[0]), ll);
            const uint32_t S0 = (sb & 0x7fffffu) | 0x800000u;
            const uint32_t S = S0 + ((S0 & 1u) ? rf1s : rf0s);
            if (S < (1u << 24)) { /* the functions do not decrease: no block of the run left the binade either */
                sb = (sb & 0xff800000u) | (S & 0x7fffffu);
                blk = last;
                continue;
            }
        }
        if (fast) {
            const uint32_t bf0 = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? rf0[1] : rf0[0]), l);
            const uint32_t bf1 = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? rf1[1] : rf1[0]), l);
            const uint32_t S0 = (sb & 0x7fffffu) | 0x800000u;
            const uint32_t S = S0 + ((S0 & 1u) ? bf1 : bf0);
            if (S < (1u << 24))
                sb = (sb & 0xff800000u) | (S & 0x7fffffu);
            else
                fast = false; /* left the binade after all */
        }
        if (fast)
            continue;
        /* wave-uniform from here */
        if (blk == w_blk) {
            /* sub-block by sub-block with the totals the functions kernel left (lane 16 c + j: exponent ca + c, sub-block j) */
            const int ca = __builtin_amdgcn_readlane(hi ? rca[1] : rca[0], l);
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? rslot[1] : rslot[0]), l);
            const uint32_t sub = slow_sub[wave][slot][lane]; /* lane 32 c + 16 p + j: exponent ca + c, S even / odd, sub-block j */
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = ((int)(sb >> 23) - 127) - ca;
                bool ok = false;
                if (c == 0 || c == 1) {
                    const uint32_t S0 = (sb & 0x7fffffu) | 0x800000u;
                    const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)sub, 32 * c + 16 * (int)(S0 & 1u) + j);
                    if (S0 + tj < (1u << 24)) {
                        sb = (sb & 0xff800000u) | ((S0 + tj) & 0x7fffffu);
                        ok = true;
                    }
                }
                if (!ok) { /* the sum leaves its binade in here: 64 additions, in order */
                    const uint32_t g = blk * (uint32_t)FS_BLOCK + 64u * j + (uint32_t)lane;
                    const float x = g < n ? fm_word_value<FMT>(w[j], inv, wave) : 0.0f;
                    float sum = __uint_as_float(sb);
                    sum = fsum_lanes_in_order(sum, x, seq_vals[wave], lane);
                    sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum));
                }
            }
            w_blk = next_walk(blk + 1u);
            if (w_blk < nblk)
                fetch(w_blk);
        } else { /* no sub-block totals (a misprediction, a sum that is still tiny): the whole block sample by sample */
            fetch(blk);
            float sum = __uint_as_float(sb);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t g = blk * (uint32_t)FS_BLOCK + 64u * j + (uint32_t)lane;
                const float x = g < n ? fm_word_value<FMT>(w[j], inv, wave) : 0.0f;
                sum = fsum_lanes_in_order(sum, x, seq_vals[wave], lane);
            }
            sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum));
            if (w_blk < nblk)
                fetch(w_blk); /* the registers held the next walked block's samples */
        }
    }
    if (lane == 0)
        out[2 * b + wave] = __uint_as_float(sb);
}

/* --dcfilter: the "generic" converters (convert.c:113-163 UC8, :165-213 SC16, :374-423 SC16Q11).
 * Per channel z = f * dc_a + z * dc_b runs through the WHOLE stream (the converter state survives
 * the calls, convert.c:476-477), and a float recurrence cannot be re-associated bit-exactly, so this is
 * one dependent chain per channel: two dependent instructions per sample, 4.2 cycles each for a lone wavefront
 * (scripts/micro/dep_chain.hip), plus the LDS traffic that feeds them: 130 Msamples/s measured (round 4; 65 before) -- 50x
 * real time for one receiver, and what this option costs.  One workgroup walks the batch in blocks:
 *   wavefronts 1-3  block k+1: samples -> f * dc_a for both channels, into LDS
 *   wavefront 0     block k:   lanes 0 (I) and 1 (Q) run the two chains in lock step, z into LDS
 *   wavefronts 1-3  block k-1: f - z, clamp, sqrt -> u16 magnitudes (what the scan kernel then reads as
 *                              MSD_FMT_MAG16) and the f32 squares for the per-buffer sums
 * (sum_level / sum_power restart with every buffer: msd_float_means_kernel<MSD_FMT_MAGSQ>). */
constexpr int DC_BLK = 1920, DC_THREADS = 1024, DC_WORKERS = DC_THREADS - 64; /* (1920 / 1024: no faster, the chain is what takes the time) */

template <int FMT>
__device__ __forceinline__ void dc_sample(const uint8_t *iq, uint64_t g, float &fi, float &fq)
{
    if (FMT == MSD_FMT_UC8) {
        const uint32_t pair = reinterpret_cast<const uint16_t *>(iq)[g];
        fi = ((float)(pair & 0xffu) - 127.5f) / 127.5f; /* convert.c:133-134: a real division */
        fq = ((float)(pair >> 8) - 127.5f) / 127.5f;
    } else {
        const uint32_t w = reinterpret_cast<const uint32_t *>(iq)[g];
        const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f); /* exact: powers of two */
        fi = (float)(int)(int16_t)(w & 0xffffu) * inv;
        fq = (float)(int)(int16_t)(w >> 16) * inv;
    }
}

template <int FMT>
__global__ void __launch_bounds__(DC_THREADS) msd_dcfilter_kernel(const uint8_t *iq, uint64_t nsamples, float dc_a,
                                                                  float dc_b, float *state /* z1_I, z1_Q */,
                                                                  uint16_t *mag, float *magsq_out, const uint32_t *skip_if)
{
    float z_in[2] = {0.0f, 0.0f};
    bool resumed = false;
    if (skip_if) { /* the parallel-in-time kernels in front (msd_dc_kernels.hip): word 0 = they did the batch; else words 16-19 =
                      the sample from which this kernel has to go on, and the two states in front of it (DcpCtl) */
        if (skip_if[0])
            return;
        const uint64_t s0 = (uint64_t)skip_if[16] | ((uint64_t)skip_if[17] << 32);
        if (s0 >= nsamples) { /* (cannot be: an unfinished batch has an unfinished block) */
            if (threadIdx.x < 2)
                state[threadIdx.x] = __uint_as_float(skip_if[18 + threadIdx.x]);
            return;
        }
        z_in[0] = __uint_as_float(skip_if[18]);
        z_in[1] = __uint_as_float(skip_if[19]);
        resumed = true;
        iq += s0 * (FMT == MSD_FMT_UC8 ? 2u : 4u);
        mag += s0;
        magsq_out += s0;
        nsamples -= s0;
    }
    __shared__ __attribute__((aligned(16))) float tv[2][2][DC_BLK + 16]; /* [block parity][channel] f * dc_a (+ 16: the chain loop's last, unused prefetch) */
    __shared__ __attribute__((aligned(16))) float zv[2][2][DC_BLK]; /* [block parity][channel] z */
    const int tid = threadIdx.x;
    const uint64_t nblk = (nsamples + DC_BLK - 1) / DC_BLK;
    float z = tid < 2 ? (resumed ? z_in[tid] : state[tid]) : 0.0f;
    for (uint64_t k = 0; k < nblk + 2; ++k) {
        if (tid >= 64) {
            const int p = tid - 64;
            if (k < nblk) { /* block k */
                const uint64_t base = k * DC_BLK;
                for (int i = p; i < DC_BLK; i += DC_WORKERS) {
                    float fi = 0.0f, fq = 0.0f;
                    if (base + i < nsamples)
                        dc_sample<FMT>(iq, base + i, fi, fq);
                    tv[k & 1][0][i] = fi * dc_a;
                    tv[k & 1][1][i] = fq * dc_a;
                }
            }
            if (k >= 2) { /* block k - 2 */
                const uint64_t base = (k - 2) * DC_BLK;
                for (int i = p; i < DC_BLK && base + i < nsamples; i += DC_WORKERS) {
                    float fi, fq;
                    dc_sample<FMT>(iq, base + i, fi, fq);
                    fi -= zv[k & 1][0][i];
                    fq -= zv[k & 1][1][i];
                    const float sq_i = fi * fi, sq_q = fq * fq;
                    float magsq = sq_i + sq_q;
                    magsq = fminf(magsq, 1.0f); /* convert.c: if (magsq > 1) magsq = 1 -- the sum of two squares is not a NaN */
                    const float m = __builtin_sqrtf(magsq);
                    mag[base + i] = (uint16_t)(m * 65535.0f + 0.5f);
                    magsq_out[base + i] = magsq;
                }
            }
        } else if (tid < 2 && k >= 1 && k <= nblk) { /* block k - 1: the two chains */
            const uint64_t base = (k - 1) * DC_BLK;
            const uint32_t cnt = nsamples - base < (uint64_t)DC_BLK ? (uint32_t)(nsamples - base) : (uint32_t)DC_BLK;
            /* z = t + z * dc_b as two plain instructions (separately rounded product and sum, like the
             * reference's SSE2 build; nothing for the compiler to contract or reorder).  Whole blocks of 32 samples go
             * through the hand-scheduled loop of msd_dc_chain_asm.h (round 4: the compiler's schedule of the same loop
             * waited for an LDS access after every fourth step, 11-13 ns per sample against 7-8), the rest one by one. */
#define DC_STEP(T, OUT)                                                                                      \
    asm volatile("v_mul_f32 %0, %1, %3\n\tv_add_f32 %0, %2, %0" : "=&v"(OUT) : "v"(z), "v"(T), "v"(dc_b)); \
    z = OUT; /* a rename, not a move: the next step reads OUT's register */
            const float *tt = tv[(k - 1) & 1][tid];
            float *zz = zv[(k - 1) & 1][tid];
            uint32_t done = 0;
            if (cnt >= 32u) {
                uint32_t ta = (uint32_t)(uintptr_t)tt, za = (uint32_t)(uintptr_t)zz;
                uint32_t nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cnt / 32u));
                done = 32u * (cnt / 32u);
                MSD_DC_CHAIN_ASM(z, dc_b, ta, za, nn);
            }
            for (uint32_t j = done; j < cnt; ++j) {
                const float t = tt[j];
                DC_STEP(t, zz[j])
            }
#undef DC_STEP
        }
        __syncthreads();
    }
    if (tid < 2)
        state[tid] = z;
}

/* ------------------------------------------------------------------------------------------ */
/* Mode A/C (demodulate2400AC, demod_2400.c:522-708)                                          */
/* ------------------------------------------------------------------------------------------ */

/* noise_level of every buffer (demod_2400.c:530-531) from the buffer sums the scan kernel left:
 *   UC8 / MAG16: integer sums -> mean_level = sum/65536/n, mean_power = sum/65535^2/n (convert.c:104-110)
 *   SC16 / SC16Q11: the sequential float sums of msd_float_means_kernel, float division by n.
 * All arithmetic is IEEE double (division and sqrt are correctly rounded on gfx950). */
__device__ inline uint32_t ac_noise_level_of(const uint64_t *sums, const float *fmeans, int use_float, uint64_t nsamples, uint32_t b)
{
    const uint64_t first = (uint64_t)b * MSD_CHUNK_SAMPLES;
    uint64_t n = nsamples > first ? nsamples - first : 0;
    if (n > MSD_CHUNK_SAMPLES)
        n = MSD_CHUNK_SAMPLES;
    double mean_level, mean_power;
    if (use_float) {
        mean_level = (double)(fmeans[2 * b] / (float)(unsigned)n);
        mean_power = (double)(fmeans[2 * b + 1] / (float)(unsigned)n);
    } else {
        mean_level = (double)sums[2 * b] / 65536.0 / (double)(unsigned)n;
        mean_power = (double)sums[2 * b + 1] / 65535.0 / 65535.0 / (double)(unsigned)n;
    }
    const double level_sq = mean_level * mean_level;
    const double noise_stddev = sqrt(mean_power - level_sq);
    const double scaled = (mean_power + noise_stddev) * 65535;
    return n ? (uint32_t)(scaled + 0.5) : 0u;
}

constexpr int ACNT = 256;                    /* threads per workgroup */

/* Every F1 position that passes all tests of demod_2400.c:581-668; the 69-sample skip-ahead after
 * a decode (:705) is left to the resolve stage.  One thread per position, magnitudes staged in
 * LDS like in the Mode S scan; output appended in position order to a workgroup-private region. */
/* All tests of demodulate2400AC (demod_2400.c:581-668) for the F1 position p of a tile whose first
 * position is sample j0 of its buffer; mags[p + 2 + d] = m[f1_sample + d]. */
template <int STAGE> /* 0: the F1 pulse test only; 1: through the F2 pulse test; 2: everything */
__device__ __forceinline__ bool ac_eval(const uint16_t *mags, int p, uint32_t j0, uint32_t mlen, uint32_t noise_level,
                                        uint32_t &f2_clock, uint32_t &modeac)
{
    const uint32_t f1_sample = j0 + (uint32_t)p;
#define ACM(x) ((uint32_t)mags[p + 2 + (int)((x) - f1_sample)])
    if (!(f1_sample >= 1 && f1_sample < mlen))
        return false;
    const uint32_t m0 = ACM(f1_sample), m1 = ACM(f1_sample + 1), m2 = ACM(f1_sample + 2);
    const uint32_t f1_level = (m0 + m1) / 2;
    if (!(ACM(f1_sample - 1) < m0 && !(m2 > m0 || m2 > m1) && !(noise_level * 2 > f1_level)))
        return false;
    if (STAGE == 0)
        return true; /* every pulse edge of a Mode S frame, and a tenth of pure noise, gets this far */
    const float f1a_power = (float)m0 * (float)m0;
    const float f1b_power = (float)m1 * (float)m1;
    const float fsum = f1a_power + f1b_power;
    const float fraction = f1b_power / fsum;
    const float frac2 = fraction * fraction;
    const float fpos = (float)f1_sample + frac2;
    const float fclk = 25.0f * fpos;
    const uint32_t f1_clock = (uint32_t)((double)fclk + 0.5);
    f2_clock = f1_clock + (87 * 14);
    const uint32_t f2_sample = f2_clock / 25;
    const uint32_t n0 = ACM(f2_sample), n1 = ACM(f2_sample + 1), n2 = ACM(f2_sample + 2);
    const uint32_t f2_level = (n0 + n1) / 2;
    if (!(ACM(f2_sample - 1) < n0 && !(n2 > n0 || n2 > n1) && !(noise_level * 2 > f2_level)))
        return false;
    if (STAGE == 1)
        return true;
    const uint32_t f1f2_level = f1_level > f2_level ? f1_level : f2_level;
    const float midpoint = __builtin_sqrtf((float)(noise_level * f1f2_level)); /* u32 product */
    const double up = (double)midpoint * 1.41421356237309504880;
    const double down = (double)midpoint / 1.41421356237309504880;
    const uint32_t signal_threshold = (uint32_t)(up + 0.5);
    const uint32_t noise_threshold = (uint32_t)(down + 0.5);
    uint32_t bits = 0, bad = 0;
    uint32_t clock = f1_clock;
    for (int bit = 0; bit < 20; ++bit, clock += 87) {
        const uint32_t s = clock / 25;
        const uint32_t x0 = ACM(s), x1 = ACM(s + 1), x2 = ACM(s + 2);
        bits <<= 1;
        if (x2 >= signal_threshold)
            bad = 1; /* noisy quiet period */
        if (x0 >= signal_threshold || x1 >= signal_threshold)
            bits |= 1;
        else if (x0 > noise_threshold && x1 > noise_threshold)
            bad = 1; /* uncertain bit */
    }
#undef ACM
    if (!((bits & 0x80020u) == 0x80020u && (bits & 0x0101Bu) == 0 && !bad))
        return false;
    /* demod_2400.c:672-685: 00 A4 A2 A1  00 B4 B2 B1  SPI C4 C2 C1  00 D4 D2 D1 */
    modeac = ((bits & 0x40000u) ? 0x0010u : 0) | ((bits & 0x20000u) ? 0x1000u : 0) |
             ((bits & 0x10000u) ? 0x0020u : 0) | ((bits & 0x08000u) ? 0x2000u : 0) |
             ((bits & 0x04000u) ? 0x0040u : 0) | ((bits & 0x02000u) ? 0x4000u : 0) |
             ((bits & 0x00800u) ? 0x0100u : 0) | ((bits & 0x00400u) ? 0x0001u : 0) |
             ((bits & 0x00200u) ? 0x0200u : 0) | ((bits & 0x00100u) ? 0x0002u : 0) |
             ((bits & 0x00080u) ? 0x0400u : 0) | ((bits & 0x00040u) ? 0x0004u : 0) |
             ((bits & 0x00004u) ? 0x0080u : 0);
    return true;
}

/* The same reply in pieces, for positions that are known to pass the F1 and F2 pulse tests (ac_eval<1>):
 * ac_head gives the clock of the first bit and the two thresholds of the bit windows (demod_2400.c:591-640),
 * ac_bit looks at one of the twenty windows (:642-668), ac_code turns the twenty bits into the reply
 * (:669-685).  Same arithmetic, operation for operation, as ac_eval<2>. */
__device__ __forceinline__ void ac_head(const uint16_t *mags, int p, uint32_t j0, uint32_t noise_level, uint32_t &f1_clock,
                                        uint32_t &signal_threshold, uint32_t &noise_threshold)
{
    const uint32_t f1_sample = j0 + (uint32_t)p;
#define ACM(x) ((uint32_t)mags[p + 2 + (int)((x) - f1_sample)])
    const uint32_t m0 = ACM(f1_sample), m1 = ACM(f1_sample + 1);
    const uint32_t f1_level = (m0 + m1) / 2;
    const float f1a_power = (float)m0 * (float)m0;
    const float f1b_power = (float)m1 * (float)m1;
    const float fsum = f1a_power + f1b_power;
    const float fraction = f1b_power / fsum;
    const float frac2 = fraction * fraction;
    const float fpos = (float)f1_sample + frac2;
    const float fclk = 25.0f * fpos;
    f1_clock = (uint32_t)((double)fclk + 0.5);
    const uint32_t f2_sample = (f1_clock + (87 * 14)) / 25;
    const uint32_t n0 = ACM(f2_sample), n1 = ACM(f2_sample + 1);
    const uint32_t f2_level = (n0 + n1) / 2;
    const uint32_t f1f2_level = f1_level > f2_level ? f1_level : f2_level;
    const float midpoint = __builtin_sqrtf((float)(noise_level * f1f2_level)); /* u32 product */
    const double up = (double)midpoint * 1.41421356237309504880;
    const double down = (double)midpoint / 1.41421356237309504880;
    signal_threshold = (uint32_t)(up + 0.5);
    noise_threshold = (uint32_t)(down + 0.5);
#undef ACM
}

/* window `bit` (0 = first) of the reply whose F1 position is p: bit 19 - bit set = a pulse, bit 31 = the window
 * spoils the reply (noisy quiet period or uncertain bit) */
__device__ __forceinline__ uint32_t ac_bit(const uint16_t *mags, int p, uint32_t j0, uint32_t f1_clock, uint32_t bit,
                                           uint32_t signal_threshold, uint32_t noise_threshold)
{
    const uint32_t f1_sample = j0 + (uint32_t)p;
    const uint32_t s = (f1_clock + 87u * bit) / 25u;
    const uint16_t *w = mags + p + 2 + (int)(s - f1_sample);
    const uint32_t x0 = w[0], x1 = w[1], x2 = w[2];
    uint32_t r = 0;
    if (x2 >= signal_threshold)
        r |= 1u << 31;
    if (x0 >= signal_threshold || x1 >= signal_threshold)
        r |= 1u << (19u - bit);
    else if (x0 > noise_threshold && x1 > noise_threshold)
        r |= 1u << 31;
    return r;
}

__device__ __forceinline__ bool ac_code(uint32_t bits /* with bit 31 = spoiled */, uint32_t &modeac)
{
    if (!((bits & 0x80020u) == 0x80020u && (bits & 0x0101Bu) == 0 && !(bits >> 31)))
        return false;
    /* demod_2400.c:672-685: 00 A4 A2 A1  00 B4 B2 B1  SPI C4 C2 C1  00 D4 D2 D1 */
    modeac = ((bits & 0x40000u) ? 0x0010u : 0) | ((bits & 0x20000u) ? 0x1000u : 0) |
             ((bits & 0x10000u) ? 0x0020u : 0) | ((bits & 0x08000u) ? 0x2000u : 0) |
             ((bits & 0x04000u) ? 0x0040u : 0) | ((bits & 0x02000u) ? 0x4000u : 0) |
             ((bits & 0x00800u) ? 0x0100u : 0) | ((bits & 0x00400u) ? 0x0001u : 0) |
             ((bits & 0x00200u) ? 0x0200u : 0) | ((bits & 0x00100u) ? 0x0002u : 0) |
             ((bits & 0x00080u) ? 0x0400u : 0) | ((bits & 0x00040u) ? 0x0004u : 0) |
             ((bits & 0x00004u) ? 0x0080u : 0);
    return true;
}

/* The Mode A/C candidate stage: every F1 position that passes all tests of demod_2400.c:581-668 (the 69-sample
 * skip-ahead after a decode, :705, is left to the resolve stage).  Wave-autonomous like the Mode S scan: every wavefront
 * owns tiles of ACW positions, its own 5 KB of LDS (magnitudes + the ordered survivor list + the scratch of the bit
 * windows) and its own slice of the output; the three ordered compactions -- F1 pulse test (a tenth of pure noise
 * passes), F2 pulse test 20.3 us later, the twenty bit windows (one position in a hundred gets there) -- are ballots and
 * DPP prefix sums inside the wavefront, so the workgroup meets once, at the very end.  (Rounds 1-3 had a 256-thread
 * workgroup per 4096-position tile with nine barriers per tile: 177 us per 128 Mi samples, of which 57 were the loads and
 * the LDS image; this one 146.) */
constexpr int ACW = 1024;                      /* F1 positions per wavefront tile: 16 per lane */
constexpr int ACW_LOAD = ACW + 72;             /* samples staged from m[j0 - 2] */
constexpr int ACW_GROUPS = ACW_LOAD / 8;       /* 137 load groups: lanes take three each (the last lanes two) */
constexpr int ACW_MAGS_BYTES = ((ACW_LOAD + 8) * 2 + 15) & ~15;
constexpr int ACW_SURV_BYTES = ACW * 2;        /* u16[ACW]: positions that pass the F1 test (then, in place, the F2 test too) */
constexpr int ACW_BITS_BYTES = 4 * 64 * 4;     /* clock, two thresholds, bits of up to 64 replies at a time */
constexpr int ACW_BYTES = ACW_MAGS_BYTES + ACW_SURV_BYTES + ACW_BITS_BYTES;
static_assert(ACW_LOAD % 8 == 0 && MSD_CHUNK_SAMPLES % ACW == 0 && ACW_BYTES % 16 == 0, "whole groups, tiles inside one buffer");

template <int FMT>
__global__ void __launch_bounds__(ACNT, 5) msd_ac_wave_kernel(const MsdScanParams P, uint32_t ntiles, uint32_t tiles_per_wave,
                                                            const uint32_t *noise_levels /* or NULL: from the sums */,
                                                            const uint64_t *sums, const float *fmeans, int use_float, msd_ac_hit *out,
                                                            uint32_t cap, msd_wg_counts *counts)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[(ACNT / 64) * ACW_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t w = blockIdx.x * (ACNT / 64) + (uint32_t)wave; /* the wavefront's region */
    unsigned char *const my = lds + wave * ACW_BYTES;
    uint16_t *const mags = reinterpret_cast<uint16_t *>(my);
    uint16_t *const surv = reinterpret_cast<uint16_t *>(my + ACW_MAGS_BYTES);
    uint32_t *const h_clk = reinterpret_cast<uint32_t *>(my + ACW_MAGS_BYTES + ACW_SURV_BYTES), *const h_sig = h_clk + 64,
                    *const h_noise = h_sig + 64, *const h_bits = h_noise + 64;
    const uint16_t *lut = P.lut;
    const uint32_t tile_lo = w * tiles_per_wave;
    uint32_t tile_hi = tile_lo + tiles_per_wave;
    if (tile_hi > ntiles)
        tile_hi = ntiles;
    uint32_t cur = 0; /* wave-uniform output cursor */
    msd_ac_hit *const mine = out + (size_t)w * cap;
    uint32_t noise_b = 0xffffffffu, noise_of_b = 0;

    constexpr int GPT_W = (ACW_GROUPS + 63) / 64;
    RawGroup<FMT> rg[GPT_W];
    uint32_t vg[GPT_W];
    /* A tile whose samples all lie inside the batch -- every tile but a batch's first and last -- is loaded from computed
     * addresses (fetch_group selects one of four sources per lane and group: a dozen 64-bit compares and selects), and in
     * the 16-bit-magnitude instantiation, the one the pipeline uses, its words go to the LDS as they came. */
    bool inside_next = false;
    auto fetch = [&](uint32_t tile) {
        const int64_t rel0 = (int64_t)((uint64_t)tile * ACW) - FRONT;
        inside_next = rel0 >= 0 && (uint64_t)rel0 + ACW_LOAD <= (P.nsamples & ~7ull); /* wave-uniform */
        if (inside_next) {
            constexpr int BPS = RawGroup<FMT>::WORDS / 2;
#pragma unroll
            for (int i = 0; i < GPT_W; ++i) {
                const int g = lane + 64 * i;
                const uint8_t *src = P.iq + (rel0 + 8 * (g < ACW_GROUPS ? g : 0)) * BPS;
                const uint4 a = *reinterpret_cast<const uint4 *>(src);
                rg[i].w[0] = a.x; rg[i].w[1] = a.y; rg[i].w[2] = a.z; rg[i].w[3] = a.w;
                if (BPS == 4) {
                    const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
                    rg[i].w[4 % RawGroup<FMT>::WORDS] = b.x; rg[i].w[5 % RawGroup<FMT>::WORDS] = b.y;
                    rg[i].w[6 % RawGroup<FMT>::WORDS] = b.z; rg[i].w[7 % RawGroup<FMT>::WORDS] = b.w;
                }
                vg[i] = 0xffu;
            }
        } else {
            const int64_t n0 = (int64_t)P.batch_first + rel0;
#pragma unroll
            for (int i = 0; i < GPT_W; ++i) {
                const int g = lane + 64 * i;
                vg[i] = fetch_group<FMT>(P, n0 + 8 * (g < ACW_GROUPS ? g : 0), rg[i]);
            }
        }
    };
    if (tile_lo < tile_hi)
        fetch(tile_lo);
    for (uint32_t tile = tile_lo; tile < tile_hi; ++tile) {
        const uint64_t pos0 = (uint64_t)tile * ACW; /* batch-relative */
        const bool inside = inside_next;
        if (FMT == MSD_FMT_MAG16 && inside) { /* wave-uniform */
#pragma unroll
            for (int i = 0; i < GPT_W; ++i) {
                const int g = lane + 64 * i;
                if (g < ACW_GROUPS)
                    *reinterpret_cast<uint4 *>(mags + 8 * g) = make_uint4(rg[i].w[0], rg[i].w[1], rg[i].w[2], rg[i].w[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < GPT_W; ++i) {
                const int g = lane + 64 * i;
                if (g < ACW_GROUPS) {
                    uint32_t mg[8];
                    convert_group<FMT>(rg[i], vg[i], lut, mg);
                    *reinterpret_cast<uint4 *>(mags + 8 * g) = pack8(mg);
                }
            }
        }
        wave_lds_sync();
        if (tile + 1 < tile_hi)
            fetch(tile + 1);
        const uint32_t b = (uint32_t)(pos0 / MSD_CHUNK_SAMPLES);
        const uint32_t j0 = (uint32_t)(pos0 % MSD_CHUNK_SAMPLES);
        const uint64_t bfirst = (uint64_t)b * MSD_CHUNK_SAMPLES;
        const uint64_t mlen64 = P.nsamples > bfirst ? P.nsamples - bfirst : 0;
        const uint32_t mlen = mlen64 > MSD_CHUNK_SAMPLES ? MSD_CHUNK_SAMPLES : (uint32_t)mlen64;
        if (b != noise_b) { /* wave-uniform */
            noise_of_b = noise_levels ? noise_levels[b] : ac_noise_level_of(sums, fmeans, use_float, P.nsamples, b);
            noise_b = b;
        }
        const uint32_t noise_level = noise_of_b;

        /* F1 test (demod_2400.c:581-589): a lane looks at its 16 consecutive positions out of a register window of 24
         * samples (three 16-byte LDS reads); m[f1_sample + d] = mags[p + 2 + d] */
        uint32_t n1;
        {
            const uint4 *w4 = reinterpret_cast<const uint4 *>(mags + 16 * lane);
            const uint4 wa = w4[0], wb = w4[1], wc = w4[2];
            const uint32_t ww[12] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y, wc.z, wc.w};
            uint32_t mask = 0;
            const uint32_t four_noise = noise_level < 0x4000u ? 4u * noise_level : 0x10000u;
            if (four_noise <= 0xffffu) { /* wave-uniform, and true unless the buffer is pure saturation */
                /* Two positions per instruction on the packed magnitudes as they come out of the LDS: with E[j] = (r[2j],
                 * r[2j+1]) and O[j] = (r[2j+1], r[2j+2]) the positions 2i and 2i+1 of the lane compare O[i] < E[i+1] (the sample
                 * in front of the pulse), E[i+2] <= min(E[i+1], O[i+1]) (the sample behind it) and, for
                 * noise_level * 2 <= (m0 + m1) / 2, which is 4 * noise_level <= m0 + m1 (the left side is even):
                 * (4 noise -. m0) -. m1 == 0 with saturating subtractions.  Position 2i ends up in bit i, 2i+1 in bit 16 + i. */
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                auto pk = [](uint32_t v) { return __builtin_bit_cast(u16x2, v); };
                auto un = [](u16x2 v) { return __builtin_bit_cast(uint32_t, v); };
                const u16x2 Q = pk(four_noise * 0x10001u), one = pk(0x10001u);
                uint32_t acc = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u16x2 prev = pk(__builtin_amdgcn_alignbit(ww[i + 1], ww[i], 16));
                    const u16x2 m0 = pk(ww[i + 1]), m1 = pk(__builtin_amdgcn_alignbit(ww[i + 2], ww[i + 1], 16)), m2 = pk(ww[i + 2]);
                    const u16x2 a = __builtin_elementwise_sub_sat(m0, prev);
                    const u16x2 bc = __builtin_elementwise_sub_sat(m2, __builtin_elementwise_min(m0, m1));
                    const u16x2 dd = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(Q, m0), m1);
                    const uint32_t good = un(__builtin_elementwise_min(a, one)), bad = un(__builtin_elementwise_min(pk(un(bc) | un(dd)), one));
                    acc |= (good & ~bad) << i;
                }
                /* interleave: bit k of mask = position k */
                uint32_t ev = acc & 0xffu, od = acc >> 16;
                ev = (ev | (ev << 4)) & 0x0f0fu; od = (od | (od << 4)) & 0x0f0fu;
                ev = (ev | (ev << 2)) & 0x3333u; od = (od | (od << 2)) & 0x3333u;
                ev = (ev | (ev << 1)) & 0x5555u; od = (od | (od << 1)) & 0x5555u;
                mask = ev | (od << 1);
                const uint32_t first = j0 + 16u * (uint32_t)lane; /* f1_sample >= 1 && f1_sample < mlen */
                if (first == 0)
                    mask &= ~1u;
                if (first + 16u > mlen)
                    mask &= first < mlen ? (1u << (mlen - first)) - 1u : 0u;
            } else {
                uint32_t r[24];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    r[2 * i] = ww[i] & 0xffffu;
                    r[2 * i + 1] = ww[i] >> 16;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t f1_sample = j0 + (uint32_t)(16 * lane + k);
                    const uint32_t m0 = r[k + 2], m1 = r[k + 3], m2 = r[k + 4];
                    const bool pass = f1_sample >= 1 && f1_sample < mlen && r[k + 1] < m0 && !(m2 > m0 || m2 > m1) &&
                                      !(noise_level * 2 > (m0 + m1) / 2);
                    mask |= (pass ? 1u : 0u) << k;
                }
            }
            const uint32_t mine_n = (uint32_t)__builtin_popcount(mask);
            const uint32_t incl = wave_incl_scan(mine_n);
            n1 = wave_last(incl);
            uint32_t idx = incl - mine_n;
            while (mask) {
                const int k = __builtin_ctz(mask);
                mask &= mask - 1;
                surv[idx++] = (uint16_t)(16 * lane + k);
            }
            wave_lds_sync();
        }
        if (MSD_AC_PRIO)
            __builtin_amdgcn_s_setprio(MSD_AC_PRIO); /* the short dependent steps behind the F1 test, as in the Mode S scan */
        /* F2 test (:591-613) of the n1 survivors, 64 at a time, compacted in place (the write index never passes the
         * read index, and a round's reads are done before its writes) */
        uint32_t n2 = 0;
        for (uint32_t r0 = 0; r0 < n1; r0 += 64) { /* wave-uniform */
            const uint32_t i = r0 + (uint32_t)lane;
            const uint32_t pp = i < n1 ? (uint32_t)surv[i] : 0u;
            uint32_t a, c;
            const bool keep = i < n1 && ac_eval<1>(mags, (int)pp, j0, mlen, noise_level, a, c);
            const unsigned long long bal = __ballot(keep);
            wave_lds_sync();
            if (keep)
                surv[n2 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)pp;
            n2 += (uint32_t)__popcll(bal);
            wave_lds_sync();
        }
        /* the twenty bit windows (:615-668) of the n2 replies that are left, 64 replies at a time: one lane per reply
         * for the clock and the thresholds, one lane per (reply, window) for the windows */
        uint32_t n3 = 0;
        for (uint32_t c0 = 0; c0 < n2; c0 += 64) { /* wave-uniform */
            const uint32_t nc = min(64u, n2 - c0);
            if ((uint32_t)lane < nc) {
                uint32_t clk, sg, nz;
                ac_head(mags, (int)surv[c0 + lane], j0, noise_level, clk, sg, nz);
                h_clk[lane] = clk;
                h_sig[lane] = sg;
                h_noise[lane] = nz;
                h_bits[lane] = 0;
            }
            wave_lds_sync();
            for (uint32_t t = (uint32_t)lane; t < nc * 20u; t += 64u) {
                const uint32_t i = t / 20u, bit = t - i * 20u;
                const uint32_t r = ac_bit(mags, (int)surv[c0 + i], j0, h_clk[i], bit, h_sig[i], h_noise[i]);
                if (r)
                    atomicOr(&h_bits[i], r);
            }
            wave_lds_sync();
            uint32_t code = 0;
            const bool keep = (uint32_t)lane < nc && ac_code(h_bits[lane], code);
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const uint32_t k = cur + n3 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (k < cap) {
                    msd_ac_hit h;
                    h.pos = pos0 + (uint64_t)surv[c0 + lane];
                    h.f2_clock = h_clk[lane] + (87 * 14);
                    h.modeac = code;
                    mine[k] = h;
                }
            }
            n3 += (uint32_t)__popcll(bal);
            wave_lds_sync();
        }
        cur += n3;
        if (MSD_AC_PRIO)
            __builtin_amdgcn_s_setprio(0);
        wave_lds_sync(); /* the next tile rewrites the magnitudes */
    }
    /* the workgroup's only meeting: every region leaves its count, where it starts inside the workgroup's output (ntries)
     * and -- in the first region's record -- the workgroup's total (pad), so that the gather kernel adds up a thousand
     * workgroup totals instead of five thousand region counts */
    __shared__ uint32_t blk_count[ACNT / 64];
    if (lane == 0)
        blk_count[wave] = cur < cap ? cur : cap;
    __syncthreads();
    if (lane == 0) {
        msd_wg_counts c;
        uint32_t base = 0, total = 0;
        for (int i = 0; i < ACNT / 64; ++i) {
            if (i < wave)
                base += blk_count[i];
            total += blk_count[i];
        }
        c.nhits = cur;
        c.ntries = base;
        c.overflow = cur > cap ? 1u : 0u;
        c.pad = total;
        counts[w] = c;
    }
}

/* dense[offset[w] + i] = region[w][i] for the Mode A/C list.  Every workgroup adds up the counts in front of its own
 * (a thousand loads: cheaper than the single-workgroup offsets kernel and the launch gap it used to wait behind), and
 * the last one, which has seen them all, leaves the batch's totals. */
__global__ void __launch_bounds__(256) msd_ac_gather_kernel(const msd_wg_counts *counts, uint32_t nblocks,
                                                            const msd_ac_hit *regions, uint32_t cap,
                                                            msd_ac_hit *dense, uint64_t dense_cap, uint64_t *totals /* [4] */)
{
    constexpr uint32_t RPB = ACNT / 64; /* regions per workgroup of msd_ac_wave_kernel */
    __shared__ unsigned long long part[4];
    __shared__ uint32_t ovf_any;
    const uint32_t blk = blockIdx.x;
    if (threadIdx.x == 0)
        ovf_any = 0;
    __syncthreads();
    unsigned long long mine = 0;
    uint32_t ovf = 0;
    const bool last = blk == nblocks - 1;
    for (uint32_t i = threadIdx.x; i < (last ? nblocks : blk); i += 256) { /* (the last one has seen them all: the overflow flag) */
        if (i < blk)
            mine += counts[i * RPB].pad;
        if (last) {
#pragma unroll
            for (uint32_t r = 0; r < RPB; ++r)
                ovf |= counts[i * RPB + r].overflow;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = mine;
    if (ovf)
        atomicOr(&ovf_any, 1u);
    __syncthreads();
    const uint64_t o = part[0] + part[1] + part[2] + part[3];
    if (last && threadIdx.x == 0) {
        totals[0] = o + counts[blk * RPB].pad;
        totals[1] = 0;
        totals[2] = ovf_any ? 1u : 0u;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(dense);
    for (uint32_t r = 0; r < RPB; ++r) {
        const msd_wg_counts c = counts[blk * RPB + r];
        const uint32_t n = c.nhits < cap ? c.nhits : cap;
        const uint4 *src = reinterpret_cast<const uint4 *>(regions + (size_t)(blk * RPB + r) * cap);
        const uint64_t at = o + c.ntries;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            if (at + i < dense_cap)
                dst[at + i] = src[i];
    }
}

} /* namespace */

/* ------------------------------------------------------------------------------------------ */
/* launchers (C linkage, used by msd_capi.cpp)                                                */
/* ------------------------------------------------------------------------------------------ */

extern "C" uint32_t msd_scan_tile(int format)
{
    return 1024u * (uint32_t)tile_runs(format);
}

extern "C" size_t msd_scan_lds_bytes(int format)
{
    return format == MSD_FMT_UC8 ? (size_t)LDS_UC8 : (size_t)LDS_COMMON;
}

template <int FMT, bool FIX2, bool EMIT>
static int launch_scan_fix(const MsdScanParams *p, uint32_t nregions, hipStream_t stream)
{
    const size_t lds = msd_scan_lds_bytes(FMT);
    const uint32_t nwg = (nregions + WAVES - 1) / WAVES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_scan_kernel<FMT, FIX2, EMIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return -5;
    hipLaunchKernelGGL((msd_scan_kernel<FMT, FIX2, EMIT>), dim3(nwg), dim3(NT), lds, stream, *p);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

template <int FMT>
static int launch_scan_fmt(const MsdScanParams *p, uint32_t nregions, hipStream_t stream)
{
    if (p->emit.nbuffers)
        return p->fix2_112 ? launch_scan_fix<FMT, true, true>(p, nregions, stream)
                           : launch_scan_fix<FMT, false, true>(p, nregions, stream);
    return p->fix2_112 ? launch_scan_fix<FMT, true, false>(p, nregions, stream)
                       : launch_scan_fix<FMT, false, false>(p, nregions, stream);
}

extern "C" int msd_launch_scan(const MsdScanParams *p, int format, uint32_t nregions, hipStream_t stream)
{
    switch (format) {
    case MSD_FMT_UC8: return launch_scan_fmt<MSD_FMT_UC8>(p, nregions, stream);
    case MSD_FMT_SC16: return launch_scan_fmt<MSD_FMT_SC16>(p, nregions, stream);
    case MSD_FMT_SC16Q11: return launch_scan_fmt<MSD_FMT_SC16Q11>(p, nregions, stream);
    case MSD_FMT_MAG16: return launch_scan_fmt<MSD_FMT_MAG16>(p, nregions, stream);
    default: return -22;
    }
}

extern "C" int msd_launch_gather(const msd_region_counts *counts, const msd_wg_totals *wg_totals, uint32_t nwg,
                                 uint64_t *totals, const msd_hit *hits,
                                 const msd_try *tries, uint32_t hcap, uint32_t tcap, msd_hit *dense_hits,
                                 uint64_t dense_hcap, msd_try *dense_tries, uint64_t dense_tcap, uint64_t *sums,
                                 uint32_t nbuffers, uint64_t *h_totals, uint64_t *h_sums, void *wipe, uint32_t wipe_bytes,
                                 const void *tail_src, void *tail_dst, uint32_t tail_bytes, uint32_t region_len,
                                 uint32_t *buf_first, uint32_t try_abs, hipStream_t stream)
{
    hipLaunchKernelGGL(msd_gather_kernel, dim3(nwg), dim3(256), 0, stream, counts, wg_totals, hits, tries, hcap, tcap, dense_hits,
                       dense_hcap, dense_tries, dense_tcap, totals, sums, nbuffers, h_totals, h_sums,
                       static_cast<uint4 *>(wipe), wipe_bytes / 16, static_cast<const uint32_t *>(tail_src),
                       static_cast<uint32_t *>(tail_dst), tail_bytes / 4, region_len, buf_first, try_abs);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_power(const MsdScanParams *p, int format, const uint64_t *d_req, uint32_t nreq,
                                unsigned long long *d_out, hipStream_t stream)
{
    if (nreq == 0)
        return 0;
    const dim3 grid((nreq + 3) / 4), block(256);
    switch (format) {
    case MSD_FMT_UC8:
        hipLaunchKernelGGL(msd_power_kernel<MSD_FMT_UC8>, grid, block, 0, stream, *p, d_req, nreq, d_out);
        break;
    case MSD_FMT_SC16:
        hipLaunchKernelGGL(msd_power_kernel<MSD_FMT_SC16>, grid, block, 0, stream, *p, d_req, nreq, d_out);
        break;
    case MSD_FMT_SC16Q11:
        hipLaunchKernelGGL(msd_power_kernel<MSD_FMT_SC16Q11>, grid, block, 0, stream, *p, d_req, nreq, d_out);
        break;
    case MSD_FMT_MAG16:
        hipLaunchKernelGGL(msd_power_kernel<MSD_FMT_MAG16>, grid, block, 0, stream, *p, d_req, nreq, d_out);
        break;
    default:
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_power_buffers(const MsdScanParams *p, int format, const msd_acc *acc, const msd_try *tries,
                                        const uint32_t *nmsgs, uint32_t nbuffers, const uint64_t *totals,
                                        unsigned long long *out, const uint32_t *nac, uint32_t *rec_off,
                                        hipStream_t stream)
{
    if (nbuffers == 0)
        return 0;
    const dim3 grid(nbuffers * PB_WGS), block(256);
    switch (format) {
    case MSD_FMT_UC8:
        hipLaunchKernelGGL(msd_power_buffers_kernel<MSD_FMT_UC8>, grid, block, 0, stream, *p, acc, tries, nmsgs, totals, out, nac, nbuffers, rec_off);
        break;
    case MSD_FMT_SC16:
        hipLaunchKernelGGL(msd_power_buffers_kernel<MSD_FMT_SC16>, grid, block, 0, stream, *p, acc, tries, nmsgs, totals, out, nac, nbuffers, rec_off);
        break;
    case MSD_FMT_SC16Q11:
        hipLaunchKernelGGL(msd_power_buffers_kernel<MSD_FMT_SC16Q11>, grid, block, 0, stream, *p, acc, tries, nmsgs, totals, out, nac, nbuffers, rec_off);
        break;
    case MSD_FMT_MAG16:
        hipLaunchKernelGGL(msd_power_buffers_kernel<MSD_FMT_MAG16>, grid, block, 0, stream, *p, acc, tries, nmsgs, totals, out, nac, nbuffers, rec_off);
        break;
    default:
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_ac(const MsdScanParams *p, int format, const uint64_t *d_sums, const float *d_fmeans,
                             uint32_t nbuffers, uint32_t *d_noise, int noise_ready, msd_ac_hit *d_regions,
                             uint64_t region_total, msd_wg_counts *d_counts, uint64_t *d_offsets,
                             uint64_t *d_totals, msd_ac_hit *d_dense, uint64_t dense_cap, uint32_t max_wg, int phase,
                             hipStream_t stream)
{
    /* phase 0: candidates and gather; 1: the candidate kernel only, 2: the gather only (of what phase 1 of the same batch
     * left in d_regions / d_counts, on any stream behind it) */
    if (nbuffers == 0)
        return 0;
    /* noise_ready 1: the caller's levels (d_noise); else the candidate kernel works them out from the buffers' sums
     * (2: from the float sums whatever the format -- magnitudes behind the DC filter) */
    const int use_float = (format == MSD_FMT_SC16 || format == MSD_FMT_SC16Q11 || noise_ready == 2);
    const uint32_t *levels = noise_ready == 1 ? d_noise : nullptr;
    (void)nbuffers;
    /* max_wg regions, one per wavefront of msd_ac_wave_kernel (ACNT / 64 wavefronts a workgroup) */
    const uint32_t ntiles = (uint32_t)((p->nsamples + ACW - 1) / ACW);
    if (ntiles == 0) {
        if (phase != 1)
            (void)hipMemsetAsync(d_totals, 0, 4 * sizeof(uint64_t), stream);
        return 0;
    }
    uint32_t tpw = (ntiles + max_wg - 1) / max_wg;
    if (tpw == 0)
        tpw = 1;
    const uint32_t nwg = (ntiles + tpw - 1) / tpw; /* regions = wavefronts */
    const uint32_t nblocks = (nwg + ACNT / 64 - 1) / (ACNT / 64);
    uint64_t cap = region_total / ((uint64_t)nblocks * (ACNT / 64));
    if (cap > (uint64_t)tpw * ACW)
        cap = (uint64_t)tpw * ACW;
#define MSD_AC_LAUNCH(F)                                                                                                     \
    hipLaunchKernelGGL(msd_ac_wave_kernel<F>, dim3(nblocks), dim3(ACNT), 0, stream, *p, ntiles, tpw, levels, d_sums, d_fmeans, \
                       use_float, d_regions, (uint32_t)cap, d_counts)
    if (phase != 2)
        switch (format) {
        case MSD_FMT_UC8: MSD_AC_LAUNCH(MSD_FMT_UC8); break;
        case MSD_FMT_SC16: MSD_AC_LAUNCH(MSD_FMT_SC16); break;
        case MSD_FMT_SC16Q11: MSD_AC_LAUNCH(MSD_FMT_SC16Q11); break;
        case MSD_FMT_MAG16: MSD_AC_LAUNCH(MSD_FMT_MAG16); break;
        default: return -22;
        }
#undef MSD_AC_LAUNCH
    (void)d_offsets;
    if (phase != 1)
        hipLaunchKernelGGL(msd_ac_gather_kernel, dim3(nblocks), dim3(256), 0, stream, d_counts, nblocks, d_regions, (uint32_t)cap, d_dense,
                       dense_cap, d_totals);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

/* convert_sc16q11_table, convert.c:297-328: |I| & 2047 and |Q| & 2047 (abs of the sample as an int: -32768 -> 32768 -> 0), their
 * top `bits` bits index the table.  Eight samples a thread; the magnitudes go to `mag`, which the scan kernel then reads as
 * MSD_FMT_MAG16 -- its integer level / power sums are this converter's (same formulas as UC8, convert.c:318-326) -- and, for
 * the converter entry, `sums` gets them here. */
template <bool LDS_TABLE>
__global__ void __launch_bounds__(LDS_TABLE ? 1024 : 256) msd_q11_table_kernel(const uint8_t *iq, uint64_t nsamples, const uint16_t *table,
                                                                              int bits, uint16_t *mag, unsigned long long *sums /* [2] or NULL */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char q11_lds[];
    const uint16_t *tab = table;
    if (LDS_TABLE) { /* up to eight bits the table fits the LDS (128 KB): one copy per compute unit, gathers at LDS speed */
        const uint32_t words = (1u << (2 * bits)) * 2u / 16u;
        const uint4 *g = reinterpret_cast<const uint4 *>(table);
        uint4 *l = reinterpret_cast<uint4 *>(q11_lds);
        for (uint32_t i = threadIdx.x; i < (words ? words : 1u); i += blockDim.x)
            if (words)
                l[i] = g[i];
            else if (i == 0)
                *reinterpret_cast<uint2 *>(q11_lds) = *reinterpret_cast<const uint2 *>(table); /* one bit: four entries */
        __syncthreads();
        tab = reinterpret_cast<const uint16_t *>(q11_lds);
    }
    const int lose = 11 - bits;
    const uint64_t ngroups = (nsamples + 7) / 8;
    unsigned long long sl = 0, sp = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + 8 * g;
        if (8 * g + 8 <= nsamples) {
            const uint4 a = *reinterpret_cast<const uint4 *>(src), b = *reinterpret_cast<const uint4 *>(src + 4);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        } else {
            for (int k = 0; k < 8; ++k)
                if (8 * g + k < nsamples)
                    w[k] = src[k];
        }
        uint32_t m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int I = (int)(int16_t)(w[k] & 0xffffu), Q = (int)(int16_t)(w[k] >> 16);
            const uint32_t ai = (uint32_t)(I < 0 ? -I : I) & 2047u, aq = (uint32_t)(Q < 0 ? -Q : Q) & 2047u;
            m[k] = 8 * g + k < nsamples ? tab[((ai >> lose) << bits) | (aq >> lose)] : 0u;
            sl += m[k];
            sp += (unsigned long long)(m[k] * m[k]);
        }
        const uint4 packed = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
        if (8 * g + 8 <= nsamples) {
            *reinterpret_cast<uint4 *>(mag + 8 * g) = packed;
        } else {
            for (int k = 0; k < 8; ++k)
                if (8 * g + k < nsamples)
                    mag[8 * g + k] = (uint16_t)m[k];
        }
    }
    if (sums) {
        for (int o = 32; o > 0; o >>= 1) {
            sl += __shfl_down(sl, o);
            sp += __shfl_down(sp, o);
        }
        if ((threadIdx.x & 63) == 0 && (sl | sp)) {
            atomicAdd(&sums[0], sl);
            atomicAdd(&sums[1], sp);
        }
    }
}

extern "C" int msd_launch_q11_table(const void *d_iq, uint64_t nsamples, const uint16_t *d_table, int bits, uint16_t *d_mag,
                                    unsigned long long *d_sums, int cu_count, hipStream_t stream)
{
    if (!nsamples)
        return 0;
    if (bits < 1 || bits > 11)
        return -22;
    const uint64_t ngroups = (nsamples + 7) / 8;
    /* the table in the LDS (one copy per compute unit, up to 128 KB at 8 bits) where the device grants that much to a
     * workgroup; otherwise -- a smaller or a shared part, or the attribute call refused -- the instantiation that reads it
     * through the caches.  cu_count is the context's (the device the stream belongs to), not a process-wide guess. */
    if (bits <= 8 && ngroups >= 4096 && cu_count > 0) {
        const size_t lds = ((size_t)2 << (2 * bits)) < 16 ? 16 : ((size_t)2 << (2 * bits));
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_q11_table_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) == hipSuccess) {
            uint64_t blocks = (ngroups + 1023) / 1024;
            if (blocks > (uint64_t)cu_count)
                blocks = (uint64_t)cu_count;
            hipLaunchKernelGGL(msd_q11_table_kernel<true>, dim3((uint32_t)blocks), dim3(1024), lds, stream, static_cast<const uint8_t *>(d_iq),
                               nsamples, d_table, bits, d_mag, d_sums);
            return hipGetLastError() == hipSuccess ? 0 : -5;
        }
        (void)hipGetLastError(); /* refused: not an error of this launch */
    }
    uint64_t blocks = (ngroups + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(msd_q11_table_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, stream, static_cast<const uint8_t *>(d_iq), nsamples,
                       d_table, bits, d_mag, d_sums);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_convert(int format, const void *d_iq, uint32_t nsamples, const uint16_t *d_lut,
                                  uint16_t *d_mag, unsigned long long *d_sums, hipStream_t stream)
{
    const uint32_t ngroups = (nsamples + 7) / 8;
    uint32_t grid = (ngroups + 255) / 256;
    if (grid > 2048)
        grid = 2048;
    if (grid == 0)
        grid = 1;
    const uint8_t *iq = static_cast<const uint8_t *>(d_iq);
    switch (format) {
    case MSD_FMT_UC8:
        hipLaunchKernelGGL(msd_convert_kernel<MSD_FMT_UC8>, dim3(grid), dim3(256), 0, stream, iq, nsamples,
                           d_lut, d_mag, d_sums);
        break;
    case MSD_FMT_SC16:
        hipLaunchKernelGGL(msd_convert_kernel<MSD_FMT_SC16>, dim3(grid), dim3(256), 0, stream, iq, nsamples,
                           d_lut, d_mag, d_sums);
        break;
    case MSD_FMT_SC16Q11:
        hipLaunchKernelGGL(msd_convert_kernel<MSD_FMT_SC16Q11>, dim3(grid), dim3(256), 0, stream, iq,
                           nsamples, d_lut, d_mag, d_sums);
        break;
    default:
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_dcfilter(int format, const void *d_iq, uint64_t nsamples, float dc_a, float dc_b,
                                   float *d_state, uint16_t *d_mag, float *d_magsq, const void *d_skip_if, hipStream_t stream)
{
    const uint32_t *skip_if = static_cast<const uint32_t *>(d_skip_if);
    const uint8_t *iq = static_cast<const uint8_t *>(d_iq);
    if (nsamples == 0)
        return 0;
    switch (format) {
    case MSD_FMT_UC8:
        hipLaunchKernelGGL(msd_dcfilter_kernel<MSD_FMT_UC8>, dim3(1), dim3(DC_THREADS), 0, stream, iq, nsamples, dc_a,
                           dc_b, d_state, d_mag, d_magsq, skip_if);
        break;
    case MSD_FMT_SC16:
        hipLaunchKernelGGL(msd_dcfilter_kernel<MSD_FMT_SC16>, dim3(1), dim3(DC_THREADS), 0, stream, iq, nsamples, dc_a,
                           dc_b, d_state, d_mag, d_magsq, skip_if);
        break;
    case MSD_FMT_SC16Q11:
        hipLaunchKernelGGL(msd_dcfilter_kernel<MSD_FMT_SC16Q11>, dim3(1), dim3(DC_THREADS), 0, stream, iq, nsamples,
                           dc_a, dc_b, d_state, d_mag, d_magsq, skip_if);
        break;
    default:
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" size_t msd_fm_work_bytes(uint32_t nbuffers)
{
    return sizeof(FmBufWork) * (size_t)(nbuffers ? nbuffers : 1u);
}

extern "C" int msd_launch_dc_sums(const float *d_magsq, uint64_t nsamples, uint64_t buffer_len, uint32_t nbuffers,
                                  float *d_out, void *d_work, int phase, hipStream_t stream)
{
    return msd_launch_float_means(MSD_FMT_MAGSQ, d_magsq, nsamples, buffer_len, nbuffers, d_out, nullptr, d_work, phase, stream);
}

template <int FMT>
static void launch_fm(const uint8_t *iq, uint64_t nsamples, uint64_t buffer_len, uint32_t nbuffers, float *d_out,
                      const float *tile_sums, FmBufWork *work, int phase, hipStream_t stream)
{
    if (phase != 2) {
        if (!tile_sums)
            hipLaunchKernelGGL(msd_fm_totals_kernel<FMT>, dim3(nbuffers * FM_PARTS), dim3(FMK_THREADS), 0, stream, iq, nsamples,
                               buffer_len, nbuffers, work);
        hipLaunchKernelGGL(msd_fm_functions_kernel<FMT>, dim3(nbuffers * FM_PARTS), dim3(FMK_THREADS), 0, stream, iq, nsamples,
                           buffer_len, nbuffers, work, tile_sums);
    }
    if (phase != 1)
        hipLaunchKernelGGL(msd_fm_apply_kernel<FMT>, dim3(nbuffers), dim3(128), 0, stream, iq, nsamples, buffer_len, nbuffers,
                           static_cast<const FmBufWork *>(work), d_out);
}

/* phase 0: everything; 1: the block functions only, 2: the apply walk only (what phase 1 left in d_work, on any stream
 * behind it).  msd_fm_deferrable(): whether the two phases exist for this call (else phase 1 does everything, 2 nothing). */
extern "C" int msd_fm_deferrable(const void *d_work, uint64_t buffer_len, uint32_t nbuffers)
{
    return d_work && nbuffers && buffer_len <= (uint64_t)FB_MAX * FS_BLOCK;
}

extern "C" int msd_launch_float_means(int format, const void *d_iq, uint64_t nsamples, uint64_t buffer_len,
                                      uint32_t nbuffers, float *d_out, const float *tile_sums, void *d_work, int phase,
                                      hipStream_t stream)
{
    if (buffer_len % FS_BLOCK)
        tile_sums = nullptr;
    const uint8_t *iq = static_cast<const uint8_t *>(d_iq);
    const uint32_t grid = nbuffers; /* one workgroup per buffer */
    if (msd_fm_deferrable(d_work, buffer_len, nbuffers)) {
        FmBufWork *work = static_cast<FmBufWork *>(d_work);
        if (format == MSD_FMT_SC16)
            launch_fm<MSD_FMT_SC16>(iq, nsamples, buffer_len, nbuffers, d_out, tile_sums, work, phase, stream);
        else if (format == MSD_FMT_SC16Q11)
            launch_fm<MSD_FMT_SC16Q11>(iq, nsamples, buffer_len, nbuffers, d_out, tile_sums, work, phase, stream);
        else if (format == MSD_FMT_MAGSQ) /* msd_launch_dc_sums */
            launch_fm<MSD_FMT_MAGSQ>(iq, nsamples, buffer_len, nbuffers, d_out, nullptr, work, phase, stream);
        else
            return -22;
        return hipGetLastError() == hipSuccess ? 0 : -5;
    }
    if (phase == 2)
        return 0;
    /* a buffer longer than FB_MAX blocks (the converter entry takes any length): one wavefront per sum */
    if (format == MSD_FMT_SC16)
        hipLaunchKernelGGL(msd_float_means_kernel<MSD_FMT_SC16>, dim3(grid), dim3(FM_THREADS), 0, stream, iq,
                           nsamples, buffer_len, nbuffers, d_out);
    else if (format == MSD_FMT_SC16Q11)
        hipLaunchKernelGGL(msd_float_means_kernel<MSD_FMT_SC16Q11>, dim3(grid), dim3(FM_THREADS), 0, stream, iq,
                           nsamples, buffer_len, nbuffers, d_out);
    else if (format == MSD_FMT_MAGSQ) /* msd_launch_dc_sums */
        hipLaunchKernelGGL(msd_float_means_kernel<MSD_FMT_MAGSQ>, dim3(grid), dim3(FM_THREADS), 0, stream, iq,
                           nsamples, buffer_len, nbuffers, d_out);
    else
        return -22;
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
