/*
 * msd_kernels.hip -- CDNA4 (gfx950) kernels of the Mode S / Mode A/C candidate stage.
 *
 * msd_scan_kernel fuses, per tile of 4096 scan positions:
 *   IQ -> u16 magnitude            convert.c:63-111 (UC8 table), :215-253 / :332-370 (float)
 *   preamble pre-check + 3 tests   demod_2400.c:276-330
 *   5-phase PPM bit slicing        demod_2400.c:73-229 (closed form t = 95 + tp + 12k)
 *   CRC-24 + syndrome lookup       crc.c:67-82, :389-412
 *   state-free part of scoring     mode_s.c:311-409
 *   signal power                   demod_2400.c:386-399
 * Magnitudes live only in LDS: HBM sees the IQ bytes once (+7 % halo) and a few bytes of
 * candidate records per thousand samples.  No MFMA: there is no dense contraction on this path.
 *
 * Work decomposition: persistent workgroups, each owning a contiguous run of tiles, so its
 * candidate records come out already ordered and a workgroup-private cursor replaces global
 * atomics; a prefix + gather pass (msd_offsets_kernel / msd_gather_kernel) then concatenates the
 * per-workgroup regions into the dense, position-ordered lists the resolve stage walks.
 */
#include <hip/hip_runtime.h>

#include "msd_internal.h"
#include "msd_kernels.h"

/* The float converters must round like the reference's x86-64 build: separate multiply and add
 * (no FMA contraction) and a correctly rounded square root.  The file is compiled with
 * -ffp-contract=off as well; sqrtf is IEEE-exact under hipcc's default
 * -fhip-fp32-correctly-rounded-divide-sqrt (the __f*_rn / __fsqrt_rn intrinsics are NOT: they map
 * to plain operators and the native sqrt). */
#pragma clang fp contract(off)

namespace {

constexpr int NT = MSD_SCAN_THREADS;       /* threads per workgroup */
constexpr int T = MSD_TILE;                /* scan positions per tile */
constexpr int LOADN = MSD_TILE_LOAD;       /* samples staged per tile */
constexpr int NGROUP = LOADN / 8;          /* 8-sample load groups per tile */
constexpr int FRONT = MSD_HALO_FRONT;      /* 328 */
constexpr int HPASS = 64;                  /* hits handled per slicing pass */
constexpr int TSLOTS = HPASS * 5;          /* try slots per pass: 5 phases per hit */
constexpr int SPT = (TSLOTS + NT - 1) / NT; /* try slots per thread in blocked order */
constexpr int LUT_STRIDE = MSD_LUT_STRIDE;

static_assert(T == NT * 16, "each thread owns 16 scan positions of a tile");
static_assert(LOADN % 8 == 0 && FRONT % 8 == 0, "load groups are 8 samples");

/* ---- dynamic LDS carve-up (all offsets multiples of 16) ---- */
constexpr int OFF_MAGS = 0;                                  /* u16[LOADN + 8] */
constexpr int OFF_MASK = OFF_MAGS + (LOADN + 8) * 2;         /* u8[T] */
constexpr int OFF_CRC = OFF_MASK + T;                        /* u32[256] */
constexpr int OFF_SYN = OFF_CRC + 1024;                      /* u32[51 + 107 (+2 pad)] */
constexpr int OFF_HITS = OFF_SYN + 640;                      /* u32[HPASS] */
constexpr int OFF_NLIVE = OFF_HITS + HPASS * 4;              /* u32[HPASS] */
constexpr int OFF_TMSG = OFF_NLIVE + HPASS * 4;              /* u8[TSLOTS][16] */
constexpr int OFF_TADDR = OFF_TMSG + TSLOTS * 16;            /* u32[TSLOTS] */
constexpr int OFF_TCRC = OFF_TADDR + TSLOTS * 4;             /* u32[TSLOTS] */
constexpr int OFF_TNB = OFF_TCRC + TSLOTS * 4;               /* u8[TSLOTS]: nbytes | 0x80 live */
constexpr int OFF_TERR = OFF_TNB + TSLOTS;                   /* u8[TSLOTS] */
constexpr int OFF_SURV = OFF_TERR + TSLOTS;                  /* u16[TSLOTS] */
constexpr int OFF_LIVE = OFF_SURV + TSLOTS * 2;              /* u16[TSLOTS] */
constexpr int OFF_POWER = OFF_LIVE + TSLOTS * 2;             /* u64[TSLOTS] */
constexpr int OFF_SCR = OFF_POWER + TSLOTS * 8;              /* u32[16] scan scratch */
constexpr int OFF_LUT = OFF_SCR + 64;                        /* u16[128 * LUT_STRIDE], UC8 only */
constexpr int LDS_COMMON = OFF_LUT;
constexpr int LDS_UC8 = OFF_LUT + 128 * LUT_STRIDE * 2;
static_assert(OFF_MASK % 16 == 0 && OFF_CRC % 16 == 0 && OFF_TMSG % 16 == 0 && OFF_POWER % 16 == 0 &&
              OFF_LUT % 16 == 0, "LDS carve offsets must stay 16-byte aligned");

/* exclusive prefix sum of one value per thread over the workgroup; *total = sum */
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *scratch)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o)
            x += y;
    }
    if (lane == 63)
        scratch[w] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        uint32_t s = scratch[i];
        if (i < w)
            base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

/* (b - 127.5)^2 only depends on k = b-128 (b >= 128) or 127-b (b < 128) */
__device__ __forceinline__ uint32_t fold8(uint32_t b)
{
    return (b ^ ((b >> 7) - 1u)) & 0x7fu;
}

/* convert.c:215-253 / :332-370 float path */
__device__ __forceinline__ uint32_t mag_from_s16(int I, int Q, float inv_scale)
{
    const float fi = (float)I * inv_scale; /* division by a power of two is exact */
    const float fq = (float)Q * inv_scale;
    const float sq_i = fi * fi, sq_q = fq * fq;
    float magsq = sq_i + sq_q;
    if (magsq > 1.0f)
        magsq = 1.0f;
    const float m = __builtin_sqrtf(magsq);
    const float scaled = m * 65535.0f;
    return (uint32_t)(uint16_t)(scaled + 0.5f);
}

/* demod_2400.c:73-93 */
__device__ __forceinline__ int correlate(const uint16_t *p, int c)
{
    const int m0 = p[0], m1 = p[1], m2 = p[2];
    switch (c) {
    case 0: return 18 * m0 - 15 * m1 - 3 * m2;
    case 1: return 14 * m0 - 5 * m1 - 9 * m2;
    case 2: return 16 * m0 + 5 * m1 - 20 * m2;
    case 3: return 7 * m0 + 11 * m1 - 18 * m2;
    default: return 4 * m0 + 15 * m1 - 20 * m2 + (int)p[3];
    }
}

/* byte b of the message tried at phase tp for the preamble whose pa = &m[j]: bit k of the message
 * is correlator (95+tp+12k) % 5 at sample j + (95+tp+12k) / 5 (demod_2400.c:98-177,188-189) */
__device__ __forceinline__ uint32_t slice_byte(const uint16_t *pa, int tp, int b)
{
    uint32_t v = 0;
    int t = 95 + tp + 96 * b;
#pragma unroll
    for (int k = 0; k < 8; ++k, t += 12) {
        const int idx = t / 5, c = t - 5 * idx;
        v = (v << 1) | (correlate(pa + idx, c) > 0 ? 1u : 0u);
    }
    return v;
}

/* demod_2400.c:193-205 */
__device__ __forceinline__ int bytes_for_df(uint32_t df)
{
    /* short: 0,4,5,11  long: 16,17,18,20,21,24  else give up after one byte */
    const uint32_t short_set = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
    const uint32_t long_set = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21) | (1u << 24);
    if ((short_set >> df) & 1u)
        return 7;
    if ((long_set >> df) & 1u)
        return 14;
    return 1;
}

template <int FMT>
__device__ __forceinline__ void load_group(const MsdScanParams &P, int64_t n, uint32_t (&mg)[8],
                                           const uint16_t *lut)
{
    /* n: absolute sample index of the first of 8 samples (multiple of 8) */
    constexpr int BPS = (FMT == MSD_FMT_UC8) ? 2 : (FMT == MSD_FMT_MAG16 ? 2 : 4);
    const int64_t rel = n - (int64_t)P.batch_first;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        mg[k] = 0;
    const uint8_t *src;
    int avail = 8;
    if (rel < 0) {
        if (!P.have_prev)
            return; /* before the start of the stream / after a discontinuity: zero magnitudes */
        src = P.prev_tail + (rel + FRONT) * BPS;
    } else {
        const int64_t left = (int64_t)P.nsamples - rel;
        if (left <= 0)
            return;
        if (left < 8)
            avail = (int)left;
        src = P.iq + rel * BPS;
    }

    uint32_t w[8]; /* 32 bytes of raw input, zero padded */
#pragma unroll
    for (int k = 0; k < 8; ++k)
        w[k] = 0;
    if (avail == 8) {
        const uint4 a = *reinterpret_cast<const uint4 *>(src);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if (BPS == 4) {
            const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        }
    } else { /* ragged end of the capture */
        for (int k = 0; k < avail * BPS / 2; ++k) {
            const uint32_t h = *reinterpret_cast<const uint16_t *>(src + 2 * k);
            w[k >> 1] |= h << (16 * (k & 1));
        }
    }

    if (FMT == MSD_FMT_UC8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t pair = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu; /* I | Q << 8 */
            mg[k] = lut[fold8(pair >> 8) * LUT_STRIDE + fold8(pair & 0xffu)];
        }
    } else if (FMT == MSD_FMT_MAG16) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            mg[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
    } else {
        const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int I = (int)(int16_t)(w[k] & 0xffffu), Q = (int)(int16_t)(w[k] >> 16);
            mg[k] = mag_from_s16(I, Q, inv);
        }
    }
    if (avail < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k >= avail)
                mg[k] = 0;
    }
}

template <int FMT>
__global__ void __launch_bounds__(NT) msd_scan_kernel(const MsdScanParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *mags = reinterpret_cast<uint16_t *>(smem + OFF_MAGS);
    uint8_t *mask = smem + OFF_MASK;
    uint32_t *crc_tab = reinterpret_cast<uint32_t *>(smem + OFF_CRC);
    uint32_t *syn = reinterpret_cast<uint32_t *>(smem + OFF_SYN);
    uint32_t *hitlist = reinterpret_cast<uint32_t *>(smem + OFF_HITS);
    uint32_t *hit_nlive = reinterpret_cast<uint32_t *>(smem + OFF_NLIVE);
    uint8_t *try_msg = smem + OFF_TMSG;
    uint32_t *try_addr = reinterpret_cast<uint32_t *>(smem + OFF_TADDR);
    uint32_t *try_crc = reinterpret_cast<uint32_t *>(smem + OFF_TCRC);
    uint8_t *try_nb = smem + OFF_TNB;
    uint8_t *try_err = smem + OFF_TERR;
    uint16_t *surv = reinterpret_cast<uint16_t *>(smem + OFF_SURV);
    uint16_t *live = reinterpret_cast<uint16_t *>(smem + OFF_LIVE);
    unsigned long long *power = reinterpret_cast<unsigned long long *>(smem + OFF_POWER);
    uint32_t *scr = reinterpret_cast<uint32_t *>(smem + OFF_SCR);
    uint16_t *lut = reinterpret_cast<uint16_t *>(smem + OFF_LUT);

    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x;

    /* constant tables -> LDS, once per persistent workgroup */
    for (int i = tid; i < 256; i += NT)
        crc_tab[i] = P.crc_tab[i];
    for (int i = tid; i < 160; i += NT)
        syn[i] = (i < 51) ? (i < (int)P.nsyn56 ? P.syn56[i] : 0xffffffffu)
                          : ((i - 51) < (int)P.nsyn112 && i < 158 ? P.syn112[i - 51] : 0xffffffffu);
    if (FMT == MSD_FMT_UC8) {
        const uint4 *g = reinterpret_cast<const uint4 *>(P.lut);
        uint4 *l = reinterpret_cast<uint4 *>(lut);
        for (int i = tid; i < 128 * LUT_STRIDE * 2 / 16; i += NT)
            l[i] = g[i];
    }
    __syncthreads();

    const uint32_t tile_lo = wg * P.tiles_per_wg;
    uint32_t tile_hi = tile_lo + P.tiles_per_wg;
    if (tile_hi > P.ntiles)
        tile_hi = P.ntiles;
    const uint64_t batch_end = P.batch_first + P.nsamples; /* one past the last scan position */

    uint32_t hcur = 0, tcur = 0; /* workgroup-uniform cursors into this workgroup's regions */
    msd_hit *const my_hits = P.hits + (size_t)wg * P.hcap;
    msd_try *const my_tries = P.tries + (size_t)wg * P.tcap;

    for (uint32_t tile = tile_lo; tile < tile_hi; ++tile) {
        const uint64_t a0 = P.batch_first + (uint64_t)tile * T; /* first scan position */
        const int64_t n0 = (int64_t)a0 - FRONT;                 /* sample staged at mags[0] */

        /* ---- stage 1: IQ -> magnitudes in LDS, and this tile's share of the buffer sums ---- */
        uint32_t sum_level = 0;
        unsigned long long sum_power = 0;
        for (int g = tid; g < NGROUP; g += NT) {
            uint32_t mg[8];
            load_group<FMT>(P, n0 + 8 * g, mg, lut);
            uint4 packed;
            packed.x = mg[0] | (mg[1] << 16);
            packed.y = mg[2] | (mg[3] << 16);
            packed.z = mg[4] | (mg[5] << 16);
            packed.w = mg[6] | (mg[7] << 16);
            *reinterpret_cast<uint4 *>(mags + 8 * g) = packed;
            if (g >= FRONT / 8 && g < FRONT / 8 + T / 8) { /* samples this tile owns */
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    sum_level += mg[k];
                    sum_power += (unsigned long long)(mg[k] * mg[k]);
                }
            }
        }
        if (P.chunk_sums) {
            unsigned long long sl = sum_level;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sl += __shfl_down(sl, o);
                sum_power += __shfl_down(sum_power, o);
            }
            if ((tid & 63) == 0 && (sl | sum_power)) {
                /* owned samples are [a0, a0+T): one buffer, because T divides the buffer length */
                const uint64_t c = (a0 - P.batch_first) / MSD_CHUNK_SAMPLES;
                atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * c]), sl);
                atomicAdd(reinterpret_cast<unsigned long long *>(&P.chunk_sums[2 * c + 1]), sum_power);
            }
        }
        __syncthreads();
        if (P.debug_flags & 2)
            continue;

        /* ---- stage 2: preamble tests for every scan position (demod_2400.c:257-335) ---- */
        uint32_t nz = 0; /* positions of mine with a test that fired */
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p0 = it * (NT * 8) + tid * 8;
            uint32_t v[16];
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(mags + p0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 q = src[k];
                    v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
                }
            }
            uint32_t mlo = 0, mhi = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                /* pa[d] = mags[p + 2 + d]: the tile stages 328 samples ahead, the reference's
                 * overlap is 326 */
#define PA(d) ((int)((v[(q + 2 + (d)) >> 1] >> (16 * ((q + 2 + (d)) & 1))) & 0xffffu))
                uint32_t m = 0;
                if (PA(1) > PA(7) && PA(12) > PA(14) && PA(12) > PA(15)) {
                    const int base_noise = PA(5) + PA(8) + PA(16) + PA(17) + PA(18);
                    const int ref_level = (base_noise * P.threshold) >> 5;
                    const int diff_2_3 = PA(2) - PA(3);
                    const int sum_1_4 = PA(1) + PA(4);
                    const int diff_10_11 = PA(10) - PA(11);
                    const int common3456 = sum_1_4 - diff_2_3 + PA(9) + PA(12);
                    if (common3456 - diff_10_11 >= ref_level)
                        m |= 1u;
                    if (common3456 + diff_10_11 >= ref_level)
                        m |= 2u;
                    if (sum_1_4 + 2 * diff_2_3 + diff_10_11 + PA(12) >= ref_level)
                        m |= 4u;
                }
#undef PA
                if (a0 + (uint64_t)(p0 + q) >= batch_end)
                    m = 0; /* past the last position the reference scans */
                if (q < 4)
                    mlo |= m << (8 * q);
                else
                    mhi |= m << (8 * (q - 4));
            }
            *reinterpret_cast<uint2 *>(mask + p0) = make_uint2(mlo, mhi);
            nz |= mlo | mhi;
        }
        /* most tiles of a quiet band have no hit at all: skip the bookkeeping then */
        const int any = __syncthreads_or(nz != 0);
        if (!any || (P.debug_flags & 1))
            continue;

        /* ---- stage 3: ordered hit list ---- */
        uint32_t mymask[4];
        {
            const uint4 q = *reinterpret_cast<const uint4 *>(mask + 16 * tid);
            mymask[0] = q.x; mymask[1] = q.y; mymask[2] = q.z; mymask[3] = q.w;
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            cnt += ((mymask[k >> 2] >> (8 * (k & 3))) & 0xffu) ? 1u : 0u;
        uint32_t H;
        const uint32_t rank0 = block_excl_scan(cnt, &H, scr);

        for (uint32_t lo = 0; lo < H; lo += HPASS) {
            const uint32_t nh = (H - lo < (uint32_t)HPASS) ? (H - lo) : (uint32_t)HPASS;
            {
                uint32_t r = rank0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t m = (mymask[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    if (m) {
                        if (r >= lo && r < lo + HPASS)
                            hitlist[r - lo] = (uint32_t)(16 * tid + k) | (m << 12);
                        ++r;
                    }
                }
            }
            if (tid < HPASS)
                hit_nlive[tid] = 0;
            __syncthreads();

            /* ---- stage 4a: first byte of every tried phase -> DF -> length ---- */
            for (uint32_t s = tid; s < nh * 5; s += NT) {
                const uint32_t hit = s / 5, q = s - 5 * hit;
                const uint32_t e = hitlist[hit];
                const uint32_t m = e >> 12;
                const bool tried = (q < 2) ? (m & 1u) : ((q < 4) ? (m & 2u) : (m & 4u));
                uint32_t nb = 0;
                if (tried) {
                    const uint32_t b0 = slice_byte(mags + (e & 0xfffu) + 2, 4 + (int)q, 0);
                    nb = (uint32_t)bytes_for_df(b0 >> 3);
                    try_msg[16 * s] = (uint8_t)b0;
                }
                try_nb[s] = (uint8_t)nb;
            }
            __syncthreads();

            /* ---- stage 4b: tries with a known DF slice their remaining bytes, one lane per
             *      (try, byte) so that the short list still fills wavefronts ---- */
            uint32_t nsurv;
            {
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < SPT; ++i) {
                    const uint32_t s = tid * SPT + i;
                    if (s < nh * 5 && try_nb[s] > 1)
                        ++c;
                }
                uint32_t r = block_excl_scan(c, &nsurv, scr);
#pragma unroll
                for (int i = 0; i < SPT; ++i) {
                    const uint32_t s = tid * SPT + i;
                    if (s < nh * 5 && try_nb[s] > 1)
                        surv[r++] = (uint16_t)s;
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < nsurv * 13; i += NT) {
                const uint32_t u = i / 13, b = 1 + (i - 13 * u);
                const uint32_t s = surv[u];
                if (b < try_nb[s]) {
                    const uint32_t hit = s / 5, q = s - 5 * hit;
                    const uint32_t pos = hitlist[hit] & 0xfffu;
                    try_msg[16 * s + b] = (uint8_t)slice_byte(mags + pos + 2, 4 + (int)q, (int)b);
                }
            }
            __syncthreads();

            /* ---- stage 4c: CRC + the part of scoreModesMessage that needs no filter ---- */
            for (uint32_t u = tid; u < nsurv; u += NT) {
                const uint32_t s = surv[u];
                const uint8_t *msg = try_msg + 16 * s;
                const int n = try_nb[s];
                uint32_t orall = 0, rem = 0;
                for (int i = 0; i < n - 3; ++i) {
                    const uint32_t byte = msg[i];
                    orall |= byte;
                    rem = ((rem << 8) ^ crc_tab[byte ^ (rem >> 16)]) & 0xffffffu;
                }
                const uint32_t tail = ((uint32_t)msg[n - 3] << 16) | ((uint32_t)msg[n - 2] << 8) | msg[n - 1];
                orall |= tail;
                const uint32_t crc = rem ^ tail;
                const uint32_t df = msg[0] >> 3;
                const uint32_t aa = ((uint32_t)msg[1] << 16) | ((uint32_t)msg[2] << 8) | msg[3];
                bool alive = (orall != 0); /* mode_s.c:325 */
                uint32_t addr = crc, errbit = 0xffu;
                if (alive && (df == 11 || df == 17 || df == 18)) {
                    addr = aa;
                    const uint32_t syndrome = (df == 11) ? (crc & 0xffff80u) : crc;
                    if (syndrome != 0) {
                        /* modesChecksumDiagnose (crc.c:389-412): exact match in the sorted
                         * single-bit table, or give up */
                        const uint32_t *tab = (df == 11) ? syn : syn + 51;
                        int lo2 = 0, hi2 = (df == 11) ? (int)P.nsyn56 : (int)P.nsyn112;
                        alive = false;
                        while (lo2 < hi2) {
                            const int mid = (lo2 + hi2) >> 1;
                            const uint32_t e = tab[mid];
                            if ((e & 0xffffffu) == syndrome) {
                                errbit = e >> 24;
                                alive = true;
                                break;
                            }
                            if ((e & 0xffffffu) < syndrome)
                                lo2 = mid + 1;
                            else
                                hi2 = mid;
                        }
                        if (alive && errbit >= 8 && errbit <= 31)
                            addr ^= 1u << (31 - errbit); /* correct_aa_field, mode_s.c:266-281 */
                    }
                }
                if (alive) {
                    try_nb[s] = (uint8_t)(n | 0x80);
                    try_addr[s] = addr;
                    try_crc[s] = crc;
                    try_err[s] = (uint8_t)errbit;
                    atomicAdd(&hit_nlive[s / 5], 1u);
                }
            }
            __syncthreads();

            /* ---- stage 4d: ordered list of live tries, their signal power, and the records ---- */
            uint32_t nlive;
            {
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < SPT; ++i) {
                    const uint32_t s = tid * SPT + i;
                    if (s < nh * 5 && (try_nb[s] & 0x80))
                        ++c;
                }
                uint32_t r = block_excl_scan(c, &nlive, scr);
#pragma unroll
                for (int i = 0; i < SPT; ++i) {
                    const uint32_t s = tid * SPT + i;
                    if (s < nh * 5 && (try_nb[s] & 0x80)) {
                        live[r] = (uint16_t)s;
                        power[r] = 0;
                        ++r;
                    }
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < nlive * 17; i += NT) {
                const uint32_t l = i / 17, g = i - 17 * l;
                const uint32_t s = live[l];
                const int len = ((try_nb[s] & 0x7f) == 14) ? 268 : 134; /* msglen*12/5 */
                const uint16_t *m19 = mags + (hitlist[s / 5] & 0xfffu) + 2 + 19;
                unsigned long long acc = 0;
                const int k1 = ((int)g * 16 + 16 < len) ? (int)g * 16 + 16 : len;
                for (int k = (int)g * 16; k < k1; ++k) {
                    const uint32_t x = m19[k];
                    acc += (unsigned long long)(x * x);
                }
                if (acc)
                    atomicAdd(&power[l], acc);
            }
            __syncthreads();
            for (uint32_t l = tid; l < nlive; l += NT) {
                const uint32_t s = live[l];
                const uint32_t q = s - 5 * (s / 5);
                const uint4 *m4 = reinterpret_cast<const uint4 *>(try_msg + 16 * s);
                uint4 lo4 = *m4;
                if ((try_nb[s] & 0x7f) == 7) { /* short message: bytes 7..13 were never sliced */
                    lo4.y &= 0x00ffffffu;
                    lo4.z = 0;
                    lo4.w = 0;
                }
                /* bytes 14,15 of the first half carry tp and errbit */
                lo4.w = (lo4.w & 0xffffu) | ((4u + q) << 16) | ((uint32_t)try_err[s] << 24);
                uint4 hi4;
                hi4.x = try_addr[s];
                hi4.y = try_crc[s];
                const unsigned long long pw = power[l];
                hi4.z = (uint32_t)pw;
                hi4.w = (uint32_t)(pw >> 32);
                if (tcur + l < P.tcap) {
                    uint4 *dst = reinterpret_cast<uint4 *>(my_tries + tcur + l);
                    dst[0] = lo4;
                    dst[1] = hi4;
                }
            }
            for (uint32_t i = tid; i < nh; i += NT) {
                const uint32_t e = hitlist[i];
                const msd_hit rec = (a0 + (e & 0xfffu)) | ((msd_hit)(e >> 12) << 40) |
                                    ((msd_hit)hit_nlive[i] << 43);
                if (hcur + lo + i < P.hcap)
                    my_hits[hcur + lo + i] = rec;
            }
            tcur += nlive;
            __syncthreads();
        }
        hcur += H;
    }

    if (tid == 0) {
        msd_wg_counts c;
        c.nhits = hcur;
        c.ntries = tcur;
        c.overflow = (hcur > P.hcap || tcur > P.tcap) ? 1u : 0u;
        c.pad = 0;
        P.counts[wg] = c;
    }
}

/* exclusive offsets of the per-workgroup regions in the dense lists; single workgroup */
__global__ void __launch_bounds__(256) msd_offsets_kernel(const msd_wg_counts *counts, uint32_t nwg,
                                                          uint64_t *offsets /* [nwg][2] */,
                                                          uint64_t *totals /* [4] */)
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

/* dense[offset[w] + i] = region[w][i]; grid = nwg workgroups */
__global__ void __launch_bounds__(256) msd_gather_kernel(const msd_wg_counts *counts,
                                                         const uint64_t *offsets, const msd_hit *hits,
                                                         const msd_try *tries, uint32_t hcap,
                                                         uint32_t tcap, msd_hit *dense_hits,
                                                         uint64_t dense_hcap, msd_try *dense_tries,
                                                         uint64_t dense_tcap)
{
    const uint32_t w = blockIdx.x;
    const uint32_t nh = counts[w].nhits < hcap ? counts[w].nhits : hcap;
    const uint32_t nt = counts[w].ntries < tcap ? counts[w].ntries : tcap;
    const uint64_t ho = offsets[2 * w], to = offsets[2 * w + 1];
    const msd_hit *hs = hits + (size_t)w * hcap;
    for (uint32_t i = threadIdx.x; i < nh; i += blockDim.x)
        if (ho + i < dense_hcap)
            dense_hits[ho + i] = hs[i];
    const uint4 *ts = reinterpret_cast<const uint4 *>(tries + (size_t)w * tcap);
    uint4 *td = reinterpret_cast<uint4 *>(dense_tries);
    for (uint32_t i = threadIdx.x; i < 2 * nt; i += blockDim.x)
        if (to + (i >> 1) < dense_tcap)
            td[2 * to + i] = ts[i];
}

/* IQ -> magnitude only, for the iq_convert_fn-shaped entry point (convert.h:33-38): writes the
 * u16 magnitudes and accumulates the integer level/power sums (UC8). */
template <int FMT>
__global__ void __launch_bounds__(256) msd_convert_kernel(const uint8_t *iq, uint32_t nsamples,
                                                          const uint16_t *lut_g, uint16_t *mag,
                                                          unsigned long long *sums)
{
    __shared__ __attribute__((aligned(16))) uint16_t lut[128 * LUT_STRIDE];
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
    unsigned long long sl = 0, sp = 0;
    const uint32_t ngroups = (nsamples + 7) / 8;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        uint32_t mg[8];
        load_group<FMT>(P, (int64_t)g * 8, mg, lut);
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

/* mean_level / mean_power of the float converters are *sequential* float sums
 * (convert.c:228-252): not associative, so one thread replays each buffer in order. */
template <int FMT>
__global__ void __launch_bounds__(64) msd_float_means_kernel(const uint8_t *iq, uint64_t nsamples,
                                                             uint64_t buffer_len, uint32_t nbuffers,
                                                             float *out /* [nbuffers][2] */)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuffers)
        return;
    const uint64_t first = (uint64_t)b * buffer_len;
    uint64_t n = nsamples > first ? nsamples - first : 0;
    if (n > buffer_len)
        n = buffer_len;
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + first;
    float sum_level = 0.0f, sum_power = 0.0f;
    for (uint64_t k = 0; k < n; ++k) {
        const uint32_t w = src[k];
        const int I = (int)(int16_t)(w & 0xffffu), Q = (int)(int16_t)(w >> 16);
        const float fi = (float)I * inv, fq = (float)Q * inv;
        const float sq_i = fi * fi, sq_q = fq * fq;
        float magsq = sq_i + sq_q;
        if (magsq > 1.0f)
            magsq = 1.0f;
        const float m = __builtin_sqrtf(magsq);
        sum_power = sum_power + magsq;
        sum_level = sum_level + m;
    }
    out[2 * b] = sum_level;
    out[2 * b + 1] = sum_power;
}

} /* namespace */

/* ------------------------------------------------------------------------------------------ */
/* launchers (C linkage, used by msd_capi.cpp)                                                */
/* ------------------------------------------------------------------------------------------ */

extern "C" size_t msd_scan_lds_bytes(int format)
{
    return format == MSD_FMT_UC8 ? (size_t)LDS_UC8 : (size_t)LDS_COMMON;
}

extern "C" int msd_launch_scan(const MsdScanParams *p, int format, uint32_t nwg, hipStream_t stream)
{
    const size_t lds = msd_scan_lds_bytes(format);
    hipError_t e = hipSuccess;
    switch (format) {
    case MSD_FMT_UC8:
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_scan_kernel<MSD_FMT_UC8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(msd_scan_kernel<MSD_FMT_UC8>, dim3(nwg), dim3(NT), lds, stream, *p);
        break;
    case MSD_FMT_SC16:
        hipLaunchKernelGGL(msd_scan_kernel<MSD_FMT_SC16>, dim3(nwg), dim3(NT), lds, stream, *p);
        break;
    case MSD_FMT_SC16Q11:
        hipLaunchKernelGGL(msd_scan_kernel<MSD_FMT_SC16Q11>, dim3(nwg), dim3(NT), lds, stream, *p);
        break;
    case MSD_FMT_MAG16:
        hipLaunchKernelGGL(msd_scan_kernel<MSD_FMT_MAG16>, dim3(nwg), dim3(NT), lds, stream, *p);
        break;
    default:
        return -22;
    }
    if (e == hipSuccess)
        e = hipGetLastError();
    return e == hipSuccess ? 0 : -5;
}

extern "C" int msd_launch_gather(const msd_wg_counts *counts, uint32_t nwg, uint64_t *offsets,
                                 uint64_t *totals, const msd_hit *hits, const msd_try *tries,
                                 uint32_t hcap, uint32_t tcap, msd_hit *dense_hits,
                                 uint64_t dense_hcap, msd_try *dense_tries, uint64_t dense_tcap,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(msd_offsets_kernel, dim3(1), dim3(256), 0, stream, counts, nwg, offsets, totals);
    hipLaunchKernelGGL(msd_gather_kernel, dim3(nwg), dim3(256), 0, stream, counts, offsets, hits, tries,
                       hcap, tcap, dense_hits, dense_hcap, dense_tries, dense_tcap);
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

extern "C" int msd_launch_float_means(int format, const void *d_iq, uint64_t nsamples, uint64_t buffer_len,
                                      uint32_t nbuffers, float *d_out, hipStream_t stream)
{
    const uint8_t *iq = static_cast<const uint8_t *>(d_iq);
    const uint32_t grid = (nbuffers + 63) / 64;
    if (format == MSD_FMT_SC16)
        hipLaunchKernelGGL(msd_float_means_kernel<MSD_FMT_SC16>, dim3(grid), dim3(64), 0, stream, iq,
                           nsamples, buffer_len, nbuffers, d_out);
    else if (format == MSD_FMT_SC16Q11)
        hipLaunchKernelGGL(msd_float_means_kernel<MSD_FMT_SC16Q11>, dim3(grid), dim3(64), 0, stream, iq,
                           nsamples, buffer_len, nbuffers, d_out);
    else
        return -22;
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
