/* msd_mag_impl.h -- one magnitude of the stream by absolute sample index, from the raw IQ (convert.c): shared by the
 * kernels that look at a few samples of a batch again after its scan (signal power of the accepted messages,
 * demod_2400.c:386-399).  Device code only; the including file is compiled with -ffp-contract=off. */
#ifndef MSD_MAG_IMPL_H
#define MSD_MAG_IMPL_H

#include "modes_hip.h"
#include "msd_internal.h"

#pragma clang fp contract(off)

#ifndef MSD_SQRT_SIGNS
#define MSD_SQRT_SIGNS 1
#endif

/* (b - 127.5)^2 only depends on k = b-128 (b >= 128) or 127-b (b < 128) */
__device__ __forceinline__ uint32_t fold8(uint32_t b)
{
    return (b ^ ((b >> 7) - 1u)) & 0x7fu;
}

/* sqrtf, correctly rounded, for the values a magnitude square takes: zero or a normal float in [2^-40, 2] (the smallest
 * non-zero one is 2^-30).  The hardware's v_sqrt_f32 is within one ulp; the two neighbours are tried with an exact
 * residual each (FMA), as in the compiler's own expansion of sqrtf -- minus its rescaling of inputs below 2^-96 and
 * its special-value select, which cost 7 of its 16 instructions and cannot trigger here (scripts/micro/sqrt_check.hip
 * compares the two over every float in the range). */
__device__ __forceinline__ float msd_sqrt_cr(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const uint32_t rb = __float_as_uint(r);
    const float rm = __uint_as_float(rb - 1u), rp = __uint_as_float(rb + 1u);
#if MSD_SQRT_SIGNS
    /* Without compares and selects (4-cycle instructions each; shifts and adds issue in 2): the answer is rm, r = rm + 1 or
     * rp = rm + 2 as bit patterns, and each residual contributes its sign bit.  nm = rm r - x is negative iff r is not too
     * big (the old form's `em <= 0` chose rm; an exact zero residual is +0 under round-to-nearest, sign bit clear: rm):
     * one step up; np = rp r - x is negative iff even r is too small (the old `ep > 0`): another step up; np < 0 implies
     * nm < 0.  x = 0: r = 0, rm is the all-ones NaN, the fused multiply-add hands that NaN through with its sign bit,
     * np = +0: rm + 1 wraps to 0.  scripts/micro/sqrt_check.hip compares with the compiler's correctly rounded sqrtf over
     * zero and every float in range on the hardware; the converter tests hold zero samples and the 2^24-pair lattices. */
    const float nm = __builtin_fmaf(rm, r, -x), np = __builtin_fmaf(rp, r, -x);
    return __uint_as_float(rb - 1u + (__float_as_uint(nm) >> 31) + (__float_as_uint(np) >> 31));
#else
    const float em = __builtin_fmaf(-rm, r, x), ep = __builtin_fmaf(-rp, r, x);
    float y = em <= 0.0f ? rm : r;
    y = ep > 0.0f ? rp : y;
    return y;
#endif
}

/* convert.c:215-253 / :332-370 float path: separate multiply and add (no FMA contraction), correctly rounded sqrt */
__device__ __forceinline__ uint32_t mag_from_s16(int I, int Q, float inv_scale)
{
    const float fi = (float)I * inv_scale; /* division by a power of two is exact */
    const float fq = (float)Q * inv_scale;
    const float sq_i = fi * fi, sq_q = fq * fq;
    float magsq = sq_i + sq_q;
    magsq = fminf(magsq, 1.0f); /* convert.c: if (magsq > 1) magsq = 1 -- the sum of two squares is not a NaN */
    const float m = msd_sqrt_cr(magsq);
    const float scaled = m * 65535.0f;
    return (uint32_t)(uint16_t)(scaled + 0.5f);
}

/* Where a batch's samples are: iq[0] is absolute sample batch_first; the MSD_HALO_FRONT samples in front of it are at
 * prev_tail if have_prev; anything else (before the stream, behind a gap, past the end) is silence (fifo.c:179-182). */
struct MsdSampleSource {
    const uint8_t *iq, *prev_tail;
    int have_prev;
    uint64_t batch_first, nsamples;
};

/* Always exactly one unconditional load from a *selected* address (a sample that does not exist reads the lookup
 * table, which is always there, and is masked afterwards): a load inside a branch makes the compiler wait for it at
 * the join, and a caller that wants several samples' loads in flight would get them one round trip after the other. */
template <int FMT>
__device__ __forceinline__ uint32_t msd_stream_mag(const MsdSampleSource &S, int64_t n, const uint16_t *lut_g)
{
    constexpr int BPS = (FMT == MSD_FMT_SC16 || FMT == MSD_FMT_SC16Q11) ? 4 : 2;
    const int64_t rel = n - (int64_t)S.batch_first;
    const bool in_batch = rel >= 0 && rel < (int64_t)S.nsamples;
    const bool in_tail = rel < 0 && S.have_prev && rel >= -(int64_t)MSD_HALO_FRONT;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(lut_g);
    src = in_tail ? S.prev_tail + (rel + (int64_t)MSD_HALO_FRONT) * BPS : src;
    src = in_batch ? S.iq + rel * BPS : src;
    uint32_t m;
    if (FMT == MSD_FMT_UC8) {
        const uint32_t pair = *reinterpret_cast<const uint16_t *>(src);
        m = lut_g[fold8(pair >> 8) * MSD_LUT_STRIDE + fold8(pair & 0xffu)];
    } else if (FMT == MSD_FMT_MAG16) {
        m = *reinterpret_cast<const uint16_t *>(src);
    } else {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(src);
        const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        m = mag_from_s16((int)(int16_t)(w & 0xffffu), (int)(int16_t)(w >> 16), inv);
    }
    return (in_batch || in_tail) ? m : 0u;
}

/* The signal power of one accepted message, the whole wavefront on it: the sum of the squares of its `len` (134 or
 * 268) samples from scan position pos on (demod_2400.c:386-392).  x[] are the lane's five samples, loaded by the
 * caller (msd_power_loads) so that several messages' loads are in flight together.  Every lane returns the sum. */
template <int FMT>
__device__ __forceinline__ void msd_power_loads(const MsdSampleSource &S, const uint16_t *lut_g, uint32_t pos, uint32_t len, int lane,
                                                uint32_t (&x)[5])
{
    const int64_t n0 = (int64_t)S.batch_first + (int64_t)pos - (int64_t)MSD_OVERLAP + 19;
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const int k = lane + 64 * v;
        const uint32_t m = msd_stream_mag<FMT>(S, n0 + (k < (int)len ? k : 0), lut_g); /* unconditional, see above */
        x[v] = k < (int)len ? m : 0u;
    }
}

#endif
