// The three preamble threshold tests of demodulate2400 (demod_2400.c:281-330) as an exact integer FIR on the matrix
// pipe, priced against the scan kernel's vector-ALU form (MSD_TESTS_V2, copied here from msd_kernels.hip stage 2).
//
//   c >= (bn * thr) >> 5   <=>   32 c + 31 - thr bn >= 0   <=>   u c - w bn + (31 / g) >= 0,  g = gcd(32, thr), u = 32 / g, w = thr / g
//
// is one fixed linear form per test over the samples pa[1..18] with coefficients in {+-u, +-2u, -w}: |coefficient| <= 127
// whenever w <= 127 (every threshold up to 127, and the multiples of 2 / 4 / ... beyond: 400 = 16 * 25), so the form is a
// Toeplitz band and `v_mfma_i32_16x16x64_i8` evaluates it exactly (i32 accumulation, no rounding anywhere):
//
//   * B operand = the RAW bytes of the u16 magnitudes as they lie in the LDS (low byte, high byte, low byte, ...), one
//     ds_read_b128 per lane, each byte xor 0x80 (an unsigned byte b as the signed byte b - 128; the constant goes into
//     the accumulator's initial value).  No byte planes, no f16 conversion, no second copy in the LDS.
//   * 64 bytes of K = 32 samples; sixteen positions with taps pa[1..18] touch 33.  The window is pa[1] of the first
//     position .. pa[17] of the last; the one missing tap (pa[18] of row 15, a noise sample) is a v_mad on one of the
//     four result registers of the lanes that hold row 15.
//   * A operands: the band for the low bytes (coefficients in the even byte slots) and for the high bytes (odd slots),
//     per test: 6 x 4 registers, loop-invariant.
//   * value = 256 D_high + D_low.  Its sign is the sign of D_high + (D_low >> 8) (arithmetic shift = floor), so the low
//     product is shifted (one 2-cycle instruction per register) and handed to the high product as its C operand: the
//     matrix pipe does the addition.
//   * A column's window must start on a 16-byte boundary: the lanes' positions are 16 l + 5 + q instead of 16 l + q
//     (pa[1] = mags[p + 3]; p + 3 = 16 l + 8).  The VALU pre-check moves with them (four aligned ds_read_b128 instead
//     of five).
//   * Result layout: lane (n, g) of block b holds rows 4 g .. 4 g + 3 of column n.  With column n of block b at positions
//     256 b + 16 n + 5 .., a 4 x 4 transpose of the blocks' verdict words between the lanes n, n + 16, n + 32, n + 48
//     (v_permlane32_swap + v_permlane16_swap, two each) leaves lane l with its sixteen consecutive positions.
//
// Output: bit comparison of the three verdict planes of every position against the reference's arithmetic (host, 64-bit)
// for both forms, on noise-like, uniformly random and extreme (0 / 65535 mixes) magnitudes, thresholds 40 58 75 127 400;
// time per tile at 16 and 12 wavefronts per CU for both forms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int FRONT = 328, WT = 2048, NH = 2;
constexpr int MAGS_N = FRONT + WT + 8; /* u16 per wavefront, as in the scan kernel */
constexpr int SHIFT = 5;               /* MFMA form: lane l, run h, bit 15 - q <-> position 1024 h + 16 l + SHIFT + q */

typedef int v4i __attribute__((ext_vector_type(4)));

struct FirParams {
    int thr;      /* Modes.preambleThreshold */
    int u, w, c31; /* 32 / g, thr / g, 31 / g */
};

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ---- the scan kernel's form (msd_kernels.hip stage 2, MSD_TESTS_V2 + MSD_TESTS_SWZ), positions 1024 h + 16 l + q ---- */
__device__ __forceinline__ void tests_valu(const uint16_t *mags, int lane, int thr, uint32_t (&pl)[NH][3])
{
#pragma unroll 1
    for (int h = 0; h < NH; ++h) { /* one run after the other, as the scan kernel's register budget has them */
        uint32_t v[20];
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(mags + 1024 * h + 16 * lane);
            const bool odd_first = (lane & 16) != 0;
            const int k0 = odd_first ? 1 : 0, k2 = odd_first ? 3 : 2;
            const uint4 r0 = src[k0], r1 = src[k0 ^ 1], r2 = src[k2], r3 = src[k2 ^ 1], r4 = src[4];
            const uint4 c0 = odd_first ? r1 : r0, c1 = odd_first ? r0 : r1, c2 = odd_first ? r3 : r2, c3 = odd_first ? r2 : r3;
            v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
            v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
            v[8] = c2.x; v[9] = c2.y; v[10] = c2.z; v[11] = c2.w;
            v[12] = c3.x; v[13] = c3.y; v[14] = c3.z; v[15] = c3.w;
            v[16] = r4.x; v[17] = r4.y; v[18] = r4.z; v[19] = r4.w;
        }
        int sm[40];
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            asm("v_and_b32 %0, 0xffff, %1" : "=v"(sm[2 * k]) : "v"(v[k]));
            asm("v_lshrrev_b32 %0, 16, %1" : "=v"(sm[2 * k + 1]) : "v"(v[k]));
        }
        uint32_t p0 = 0, p1 = 0, p2 = 0;
        const int m32 = -32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#define PA(d) (sm[q + 2 + (d)])
            const int pre = (PA(7) - PA(1)) & (PA(14) - PA(12)) & (PA(15) - PA(12));
            const int base_noise = PA(5) + PA(8) + PA(16) + PA(17) + PA(18);
            int refm;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(refm) : "v"(base_noise), "v"(thr), "v"(m32));
            refm >>= 5;
            const int diff_2_3 = PA(2) - PA(3), diff_10_11 = PA(10) - PA(11);
            const int b = refm - (PA(1) + PA(4) + PA(12));
            const int r1 = b + diff_2_3 - PA(9);
            const int g0 = pre & (r1 + diff_10_11), g1 = pre & (r1 - diff_10_11);
            const int g2 = pre & (b - diff_2_3 - diff_2_3 - diff_10_11);
#undef PA
            p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)g0, 31);
            p1 = __builtin_amdgcn_alignbit(p1, (uint32_t)g1, 31);
            p2 = __builtin_amdgcn_alignbit(p2, (uint32_t)g2, 31);
        }
        pl[h][0] = p0;
        pl[h][1] = p1;
        pl[h][2] = p2;
    }
}

/* ---- the matrix-pipe form ---- */
/* minus the coefficient of pa[d] in test t (the pipe evaluates -T - 1, whose sign bit is the verdict "T >= 0") */
__device__ __host__ inline int ncoef(int t, int d, int u, int w)
{
    switch (d) {
    case 1: case 4: case 12: return -u;
    case 2: return t == 2 ? -2 * u : u;
    case 3: return t == 2 ? 2 * u : -u;
    case 9: return t == 2 ? 0 : -u;
    case 10: return t == 0 ? u : -u;
    case 11: return t == 0 ? -u : u;
    case 5: case 8: case 16: case 17: case 18: return w;
    default: return 0;
    }
}

struct FirConsts {
    v4i a[3][2];  /* [test][low / high byte band] */
    v4i cinit[3]; /* the low product's C operand: constant term, byte bias, row 15's missing tap */
    int mu;       /* row 15's missing tap: w in the lanes that hold it, 0 elsewhere */
};

__device__ __forceinline__ void fir_consts(const FirParams &F, int lane, FirConsts &C)
{
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int sum = 0;
        for (int d = 1; d <= 18; ++d)
            sum += ncoef(t, d, F.u, F.w);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            uint32_t wd[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; ++j) {
                const int kb = 16 * g + j, s = kb >> 1, plane = kb & 1, d = s - i + 1;
                const int c = (plane == v && d >= 1 && d <= 18) ? ncoef(t, d, F.u, F.w) : 0;
                wd[j >> 2] |= (uint32_t)(c & 0xff) << (8 * (j & 3));
            }
            C.a[t][v] = v4i{(int)wd[0], (int)wd[1], (int)wd[2], (int)wd[3]};
        }
        /* -T - 1 = sum ncoef (m - 32896) + 32896 sum ncoef - c31 - 1; row 15's window lacks pa[18] (ncoef = w) */
        const int k = 32896 * sum - F.c31 - 1;
        C.cinit[t] = v4i{k, k, k, g == 3 ? k - 32896 * F.w : k};
    }
    C.mu = g == 3 ? F.w : 0;
}

__device__ __forceinline__ void swap32(uint32_t &a, uint32_t &b)
{
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(uint32_t &a, uint32_t &b)
{
    asm("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

/* VAR 0: the high product takes the shifted low product as its C operand, the compiler's order; 1: the same, all three
 * low products of a block issued first, then the shifts, then the three high products, then the verdicts (the order
 * pinned with scheduling barriers); 2: low and high products independent, value = (D_high << 8) + D_low on the vector ALU */
template <bool PRE, int VAR>
__device__ __forceinline__ void tests_mfma(const uint16_t *mags, int lane, const FirConsts &C, uint32_t (&pl)[NH][3])
{
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        uint32_t X[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint16_t *col = mags + 1024 * h + 256 * b + 16 * n + 8; /* pa[1] of the column's first position */
            const uint4 raw = *reinterpret_cast<const uint4 *>(col + 8 * g);
            const int m35 = col[32];
            const v4i B = {(int)(raw.x ^ 0x80808080u), (int)(raw.y ^ 0x80808080u), (int)(raw.z ^ 0x80808080u), (int)(raw.w ^ 0x80808080u)};
            uint32_t x = 0;
            if (VAR == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    v4i lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][0], B, C.cinit[t], 0, 0, 0);
                    const int l3 = __mul24(m35, C.mu) + lo[3]; /* (not inline asm: hipcc pads no MFMA hazard in front of an asm statement) */
                    const v4i q = {lo[0] >> 8, lo[1] >> 8, lo[2] >> 8, l3 >> 8};
                    const v4i hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][1], B, q, 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        x = __builtin_amdgcn_alignbit(x, (uint32_t)hi[c], 31);
                }
            } else if (VAR == 1) {
                v4i lo[3], hi[3];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    lo[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][0], B, C.cinit[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int l3 = __mul24(m35, C.mu) + lo[t][3];
                    const v4i q = {lo[t][0] >> 8, lo[t][1] >> 8, lo[t][2] >> 8, l3 >> 8};
                    hi[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][1], B, q, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        x = __builtin_amdgcn_alignbit(x, (uint32_t)hi[t][c], 31);
            } else {
                v4i lo[3], hi[3];
                const v4i zero = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    lo[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][0], B, C.cinit[t], 0, 0, 0);
                    hi[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[t][1], B, zero, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int l3 = __mul24(m35, C.mu) + lo[t][3];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int v = (hi[t][c] << 8) + (c == 3 ? l3 : lo[t][c]);
                        x = __builtin_amdgcn_alignbit(x, (uint32_t)v, 31);
                    }
                }
            }
            X[b] = x; /* bits 11..8: test 0 of rows 4 g .. 4 g + 3, 7..4: test 1, 3..0: test 2 */
        }
        /* 4 x 4 transpose between the lanes n, n + 16, n + 32, n + 48: X[g'] <- block g of lane (n, g') */
        swap32(X[0], X[2]);
        swap32(X[1], X[3]);
        swap16(X[0], X[1]);
        swap16(X[2], X[3]);
        uint32_t p[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int sh = 8 - 4 * t;
            p[t] = (((X[0] >> sh) & 15u) << 12) | (((X[1] >> sh) & 15u) << 8) | (((X[2] >> sh) & 15u) << 4) | ((X[3] >> sh) & 15u);
        }
        if (PRE) {
            /* the pre-check (demod_2400.c:276-282) on the vector ALU, same positions: PA(d) = sm[q + d - 1] */
            const uint4 *src = reinterpret_cast<const uint4 *>(mags + 1024 * h + 16 * lane + 8);
            const bool odd_first = (lane & 16) != 0;
            const int k0 = odd_first ? 1 : 0, k2 = odd_first ? 3 : 2;
            const uint4 r0 = src[k0], r1 = src[k0 ^ 1], r2 = src[k2], r3 = src[k2 ^ 1];
            const uint4 c0 = odd_first ? r1 : r0, c1 = odd_first ? r0 : r1, c2 = odd_first ? r3 : r2, c3 = odd_first ? r2 : r3;
            const uint32_t v[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
            int sm[32];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm("v_and_b32 %0, 0xffff, %1" : "=v"(sm[2 * k]) : "v"(v[k]));
                asm("v_lshrrev_b32 %0, 16, %1" : "=v"(sm[2 * k + 1]) : "v"(v[k]));
            }
            uint32_t pp = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pre = (sm[q + 6] - sm[q]) & (sm[q + 13] - sm[q + 11]) & (sm[q + 14] - sm[q + 11]);
                pp = __builtin_amdgcn_alignbit(pp, (uint32_t)pre, 31);
            }
            p[0] &= pp;
            p[1] &= pp;
            p[2] &= pp;
        }
        pl[h][0] = p[0];
        pl[h][1] = p[1];
        pl[h][2] = p[2];
    }
}

/* MODE 0: vector ALU, 1: matrix pipe + VALU pre-check, 2: matrix pipe, tests only (no pre-check: what the pipe costs),
 * 3 / 4: mode 2 in the orders VAR 1 / 2, 5: mode 1 in the order VAR 1, 6: 24 independent MFMAs per run and nothing else
 * (the pipe's issue rate) */
template <int MODE>
__global__ void __launch_bounds__(1024) fir_kernel(const uint16_t *in, uint32_t *out, int iters, FirParams F, int waves_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_per_wg)
        return;
    uint16_t *mags = reinterpret_cast<uint16_t *>(smem) + (size_t)wave * (MAGS_N + 8);
    const size_t wv = (size_t)blockIdx.x * waves_per_wg + wave;
    const uint16_t *src = in + wv * MAGS_N;
    for (int i = lane; i < MAGS_N; i += 64)
        mags[i] = src[i];
    wave_lds_sync();
    FirConsts C;
    if (MODE != 0)
        fir_consts(F, lane, C);
    uint32_t acc[NH][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        uint32_t pl[NH][3];
        if (MODE == 0)
            tests_valu(mags, lane, F.thr, pl);
        else if (MODE == 6) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(mags + 16 * lane + 8 * (it & 7));
            const v4i B = {(int)raw.x, (int)raw.y, (int)raw.z, (int)raw.w};
            v4i d[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                d[k] = C.cinit[k % 3];
#pragma unroll
            for (int r = 0; r < 2 * 24 / 6; ++r)
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    d[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(C.a[k % 3][k / 3], B, d[k], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    pl[h][t] = (uint32_t)(d[3 * h + t][0] ^ d[3 * h + t][1] ^ d[3 * h + t][2] ^ d[3 * h + t][3]);
        } else
            tests_mfma<MODE == 1 || MODE == 5, MODE == 3 || MODE == 5 ? 1 : MODE == 4 ? 2 : 0>(mags, lane, C, pl);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc[h][t] ^= pl[h][t] + (uint32_t)it;
        /* a store behind everything the tests read, so that nothing of the loop body is loop-invariant to the compiler */
        if (lane == 0)
            mags[MAGS_N + (it & 1)] = (uint16_t)acc[0][0];
        wave_lds_sync();
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            out[((wv * NH + h) * 64 + lane) * 3 + t] = acc[h][t];
}

/* ---- host ---- */
static uint32_t rng_state = 12345;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return rng_state;
}

/* demod_2400.c:276-330 for scan position j of a wavefront's magnitudes; bit t = test t passed (pre-check included if `pre`) */
static int ref_verdicts(const uint16_t *m, int j, int thr, bool pre)
{
    const uint16_t *pa = m + j + 2; /* the scan kernel's pa[d] = mags[p + 2 + d] */
    if (pre && !(pa[1] > pa[7] && pa[12] > pa[14] && pa[12] > pa[15]))
        return 0;
    const int32_t base_noise = pa[5] + pa[8] + pa[16] + pa[17] + pa[18];
    const int32_t ref_level = (base_noise * thr) >> 5;
    const int32_t diff_2_3 = pa[2] - pa[3], sum_1_4 = pa[1] + pa[4], diff_10_11 = pa[10] - pa[11];
    const int32_t common3456 = sum_1_4 - diff_2_3 + pa[9] + pa[12];
    int r = 0;
    if (common3456 - diff_10_11 >= ref_level) r |= 1;
    if (common3456 + diff_10_11 >= ref_level) r |= 2;
    if (sum_1_4 + 2 * diff_2_3 + diff_10_11 + pa[12] >= ref_level) r |= 4;
    return r;
}

static int gcd(int a, int b) { return b ? gcd(b, a % b) : a; }

template <int MODE>
static int run(const uint16_t *d_in, uint32_t *d_out, int wgs, int waves, int iters, FirParams F, float *ms)
{
    const size_t lds = (size_t)waves * (MAGS_N + 8) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fir_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    fir_kernel<MODE><<<wgs, 1024, lds>>>(d_in, d_out, iters, F, waves);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(ms, e0, e1));
    return 0;
}

static int run_mode(int mode, const uint16_t *d_in, uint32_t *d_out, int wgs, int waves, int iters, FirParams F, float *ms)
{
    switch (mode) {
    case 0: return run<0>(d_in, d_out, wgs, waves, iters, F, ms);
    case 1: return run<1>(d_in, d_out, wgs, waves, iters, F, ms);
    case 2: return run<2>(d_in, d_out, wgs, waves, iters, F, ms);
    case 3: return run<3>(d_in, d_out, wgs, waves, iters, F, ms);
    case 4: return run<4>(d_in, d_out, wgs, waves, iters, F, ms);
    case 5: return run<5>(d_in, d_out, wgs, waves, iters, F, ms);
    default: return run<6>(d_in, d_out, wgs, waves, iters, F, ms);
    }
}
static const char *const MODE_NAMES[7] = {"vector ALU (MSD_TESTS_V2)", "matrix pipe + VALU pre-check", "matrix pipe, tests only", "tests only, phased order",
                                          "tests only, independent products", "phased order + VALU pre-check", "48 bare MFMAs per tile"};

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int timing_iters = argc > 2 ? atoi(argv[2]) : 64;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs, clockRate %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int waves = 16;
    const size_t nw = (size_t)wgs * waves;
    std::vector<uint16_t> h_in(nw * MAGS_N);
    std::vector<uint32_t> h_out(nw * NH * 64 * 3);
    uint16_t *d_in;
    uint32_t *d_out;
    CK(hipMalloc(&d_in, h_in.size() * 2));
    CK(hipMalloc(&d_out, h_out.size() * 4));

    /* ---- parity ---- */
    const int thrs[] = {40, 58, 75, 127, 400};
    const char *const dnames[] = {"noise-like (Rayleigh-ish, sigma 1500)", "uniform 0..65535", "extremes (0 / 65535 / 1 / 65534 / random)"};
    long total_bad = 0;
    for (int dk = 0; dk < 3; ++dk) {
        for (size_t i = 0; i < h_in.size(); ++i) {
            const uint32_t r = rnd();
            if (dk == 0) {
                const int a = (int)(r & 0x7ff) - 1024, b = (int)((r >> 11) & 0x7ff) - 1024;
                h_in[i] = (uint16_t)((a * a + b * b) >> 9); /* 0 .. 4096, chi-square-ish */
                if ((r >> 28) == 0)
                    h_in[i] = (uint16_t)(r >> 12); /* a loud sample now and then */
            } else if (dk == 1) {
                h_in[i] = (uint16_t)r;
            } else {
                const uint32_t s = r >> 29;
                h_in[i] = s == 0 ? 0 : s == 1 ? 65535 : s == 2 ? 1 : s == 3 ? 65534 : s == 4 ? 0 : s == 5 ? 65535 : (uint16_t)r;
            }
        }
        CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 2, hipMemcpyHostToDevice));
        for (int thr : thrs) {
            const int g = gcd(32, thr);
            FirParams F = {thr, 32 / g, thr / g, 31 / g};
            if (F.w > 127) {
                printf("thr %d: w = %d does not fit a signed byte -- vector-ALU form only\n", thr, F.w);
                continue;
            }
            for (int mode = 0; mode < 6; ++mode) {
                float ms;
                CK(hipMemset(d_out, 0, h_out.size() * 4));
                if (run_mode(mode, d_in, d_out, wgs, waves, 1, F, &ms))
                    return 1;
                CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
                long bad = 0, npos = 0, npass = 0;
                long by_q[16] = {0}, by_lg[4] = {0}, by_h[2] = {0}, by_t[3] = {0};
                const int shift = mode == 0 ? 0 : SHIFT;
                for (size_t wv = 0; wv < nw; ++wv)
                    for (int h = 0; h < NH; ++h)
                        for (int l = 0; l < 64; ++l)
                            for (int q = 0; q < 16; ++q) {
                                const int j = 1024 * h + 16 * l + shift + q;
                                const int want = ref_verdicts(h_in.data() + wv * MAGS_N, j, thr, mode < 2 || mode == 5);
                                int got = 0;
                                for (int t = 0; t < 3; ++t)
                                    got |= (int)((h_out[((wv * NH + h) * 64 + l) * 3 + t] >> (15 - q)) & 1u) << t;
                                /* (iteration 0 adds it = 0 to the planes: acc = planes) */
                                bad += got != want;
                                if (got != want) {
                                    ++by_q[q];
                                    ++by_lg[l >> 4];
                                    ++by_h[h];
                                    for (int t = 0; t < 3; ++t)
                                        by_t[t] += ((got ^ want) >> t) & 1;
                                }
                                npass += want != 0;
                                ++npos;
                            }
                printf("parity  data %-44s thr %3d (u %2d w %3d)  %-28s: %ld of %ld positions differ (%ld pass)\n", dnames[dk], thr, F.u, F.w,
                       MODE_NAMES[mode], bad, npos, npass);
                if (bad && total_bad == 0) {
                    printf("   first failing case, mismatches by q:");
                    for (int q = 0; q < 16; ++q)
                        printf(" %ld", by_q[q]);
                    printf("\n   by lane >> 4: %ld %ld %ld %ld   by run: %ld %ld   by test: %ld %ld %ld\n", by_lg[0], by_lg[1], by_lg[2], by_lg[3], by_h[0], by_h[1], by_t[0], by_t[1], by_t[2]);
                }
                total_bad += bad;
            }
        }
    }
    printf("parity total: %ld differing positions\n", total_bad);

    /* ---- timing: noise-like data, threshold 58 ---- */
    rng_state = 777;
    for (size_t i = 0; i < h_in.size(); ++i) {
        const uint32_t r = rnd();
        const int a = (int)(r & 0x7ff) - 1024, b = (int)((r >> 11) & 0x7ff) - 1024;
        h_in[i] = (uint16_t)((a * a + b * b) >> 9);
    }
    CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 2, hipMemcpyHostToDevice));
    FirParams F = {58, 16, 29, 15};
    const double clk = prop.clockRate * 1e3;
    for (int wv : {16, 12, 8, 4}) {
        for (int mode = 0; mode < 7; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                float ms;
                if (run_mode(mode, d_in, d_out, wgs, wv, timing_iters, F, &ms))
                    return 1;
                if (ms < best)
                    best = ms;
            }
            /* SIMD cycles per 64 scan positions (one position per lane): what the scan kernel's "81 issue cycles per
             * position-instruction" counts */
            const double groups_per_simd = (double)timing_iters * (WT / 64) * wv / 4.0;
            const double cyc = best * 1e-3 * clk / groups_per_simd;
            const double per_128mi = best * (134217728.0 / ((double)wgs * wv * timing_iters * WT));
            printf("timing  %2d waves/CU  %-28s: %.4f ms for %d tiles per wavefront = %.1f SIMD cycles per 64 positions; tests of a 128 Mi-sample launch on %d CUs: %.4f ms\n",
                   wv, MODE_NAMES[mode], best, timing_iters, cyc, wgs, per_128mi);
        }
    }
    return 0;
}
