/* --dcfilter, exact and parallel in time (round 6).
 *
 * The "generic" converters (convert.c:113-163 UC8, :165-213 SC16, :374-423 SC16Q11) run
 *     z = fl(fl(f * dc_a) + fl(z * dc_b))                     (convert.c:137-138)
 * per channel through the WHOLE stream.  The recurrence cannot be re-associated bit-exactly, and one dependent
 * multiply-add pair costs a lone wavefront 5.8 ns (scripts/micro/dcp_chain_occupancy.hip): 130 Msamples/s for
 * msd_dcfilter_kernel (msd_kernels.hip) with its LDS traffic, a fifth of one host core.  What the recurrence does have:
 *   (1) every step z -> fl(t + fl(z * b)) is monotone non-decreasing in z (a product with b > 0, a sum and two roundings
 *       to nearest are all monotone), hence so is the map F_i of a whole block of L samples;
 *   (2) F_i is nearly a translation with a slope just below one, so a handful of its values pin it down well.
 * So the stream is cut into blocks, and per pass
 *   msd_dcp_eval_kernel   one WAVEFRONT per (block, channel): its 64 lanes run the block's chain from 64 candidate start
 *                         states c_0 <= ... <= c_63 around the current guess S_i (S_i itself, its neighbours up to 8 units in
 *                         the last place either side, then geometrically out to whole binades: dcp_side); the samples
 *                         are wave-uniform: one coalesced load and conversion per 256 steps, the terms f * dc_a broadcast
 *                         from LDS sixteen steps ahead of the chain; it leaves the block's table prepared for the walk;
 *   msd_dcp_walk_kernel   one wavefront per channel walks the blocks in order from the last start state known EXACTLY:
 *                         Z_i = c_k for some k           -> Z_(i+1) = E_k            exactly (a table look-up)
 *                         c_k < Z_i < c_(k+1), E_k = E_(k+1) -> Z_(i+1) = E_k        exactly (monotonicity, (1))
 *                         otherwise                      -> a guess by linear interpolation between (c_k, E_k) and
 *                                                           (c_(k+1), E_(k+1)), a secant step at the right scale; the
 *                                                           walk goes on with guesses, which become the next pass's S_i.
 * The first block that had to guess is evaluated around its exact start in the next pass, so every pass extends the exact
 * prefix by at least one block; in practice the guesses are within a few units after two passes and the whole batch is exact
 * after 5-8 (12-14 for constant and alternating inputs: profiles/r06_dc_passes.txt; scripts/experiments/dc_parallel_proto.py
 * and dc_parallel_table_walk.py are the numpy prototypes).
 * Nothing is verified by comparison with a tolerance: a start state is either derived exactly or it is a guess.  A batch that
 * is not exact after the passes queued falls through to msd_dcfilter_kernel (sequential, always right) -- from the first block
 * whose end is not exact in both channels on: what the passes did derive is kept (msd_dcp_handover_kernel).
 * Afterwards msd_dcp_eval_kernel's centre lane leaves the exact state at every 64th sample and msd_dcp_out_kernel -- one LANE
 * per 64 samples -- repeats the chain from there and writes what msd_dcfilter_kernel writes: u16 magnitudes, f32 squares.
 *
 * Two ways of running the passes.  The default: msd_dcp_eval_kernel + msd_dcp_walk_kernel per pass, 24 passes queued (12 for
 * a batch of at most 128 blocks; those behind the one that finished return at once).  MSD_CFG_DC_FUSED_LAUNCH: msd_dcp_fused_kernel, ONE cooperative launch with the
 * two walking workgroups and all evaluating wavefronts resident together; an evaluating wavefront starts on pass p + 1 of its
 * block as soon as walk p has gone past it (a progress word per channel), the walk of pass p + 1 waits block by block for the
 * tables (a counter per block), so a pass costs the longer of the two instead of their sum and nothing runs once the batch is
 * exact.  Every wait is bounded: one that runs out gives the batch up to the in-order kernel.  A table that is out of date is
 * still a table of true values of the same block's map, so stale reads can cost passes, never exactness; the states at every
 * 64th sample are refreshed by a launch of their own behind it (a block is re-evaluated whenever its table's centre is not its
 * start state: `cen`).  Exact, tested -- and slower at every batch size (3.10 against 2.85 ms per 16 Mi samples, 0.91 against
 * 0.54 for one buffer, profiles/r06_dc_rate.txt): what the cooperative launch costs exceeds what the passes' launches and the
 * overlap of evaluation and walk save.  The switch stays off. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "msd_kernels.h"

namespace {

constexpr int DCP_FINE = 64; /* samples per fine block = per lane of the output kernel */

struct DcpCtl {
    uint32_t done;         /* both channels exact: the output kernel runs, the sequential kernel does not */
    uint32_t done_ch[2];
    uint32_t frontier[2];  /* first block whose start is exact and whose end is not known yet */
    uint32_t zfront[2];    /* its start state (bits) */
    uint32_t zend[2];      /* the state behind the last sample (bits), valid with done_ch */
    uint32_t passes_ch[2]; /* walks that did something (diagnostics) */
    uint32_t guessed;      /* blocks that had to guess, all passes (diagnostics) */
    uint32_t ndone;
    uint32_t prog[2];      /* fused kernel: walk w has gone past block i of channel c when prog[c] >= w (nb + 1) + i + 1 */
    uint32_t gaveup;       /* fused kernel: a bounded wait ran out, or the passes did */
    /* words 16-19, for a batch the passes did not finish (msd_dcp_handover_kernel): what they did get exact is kept -- the output
     * kernel writes the samples in front of `resume`, msd_dcfilter_kernel goes on in order from there with these states */
    uint32_t resume_lo, resume_hi; /* first sample of the first block whose end is not exact in both channels */
    uint32_t resume_z[2];          /* the two channels' states in front of it (bits) */
};
static_assert(sizeof(DcpCtl) == 20 * 4, "msd_dcfilter_kernel reads words 0 and 16-19");

/* floats in their order as integers: ord(-x) = -ord(x), ord(+-0) = 0, consecutive floats are consecutive integers */
__device__ __forceinline__ int64_t dcp_ord(uint32_t bits)
{
    const int64_t m = (int64_t)(bits & 0x7fffffffu);
    return (bits >> 31) ? -m : m;
}
__device__ __forceinline__ uint32_t dcp_unord(int64_t o)
{
    const int64_t lim = 0x7f7fffff; /* the largest finite float */
    o = o > lim ? lim : (o < -lim ? -lim : o);
    return o >= 0 ? (uint32_t)o : ((uint32_t)(-o) | 0x80000000u);
}
/* candidate k of 64, ascending, in units in the last place around the guess: every neighbour up to +-8 (where the map of
 * a block has slope one -- alternating input, no contraction at all -- only a candidate that IS the state decides exactly),
 * steps of a factor 1.5 up to 192, then a factor 4 up to 2^30 (whole binades: the first pass, whose guess is zero, is the
 * linear prediction), and 2^31 (clamped to the largest float) at the top.  The numpy prototype needed 15 passes for a
 * half-scale alternating input with plain powers of two and 6 with this set. */
__device__ const int32_t dcp_side[31] = {1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048, 4096, 16384,
                                         65536, 262144, 1048576, 4194304, 16777216, 67108864, 268435456, 1073741824};
__device__ __forceinline__ int64_t dcp_offset(int k)
{
    return k < 31 ? -(int64_t)dcp_side[30 - k] : (k == 31 ? 0 : (k == 63 ? ((int64_t)1 << 31) : (int64_t)dcp_side[k - 32]));
}

template <int FMT>
__device__ __forceinline__ float dcp_sample(const uint8_t *iq, uint64_t g, int ch)
{
    if (FMT == MSD_FMT_UC8)
        return ((float)iq[2 * g + ch] - 127.5f) / 127.5f; /* convert.c:133-134: a real division */
    const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f); /* exact: powers of two */
    return (float)(int)reinterpret_cast<const int16_t *>(iq)[2 * g + ch] * inv;
}

/* z = fl(t + fl(z * b)): two instructions, separately rounded (the file is compiled with -ffp-contract=off; the volatile asm
 * also keeps the compiler from re-associating across steps) */
#define DCP_STEP(Z, T, B) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %2, %0" : "+v"(Z) : "v"(B), "v"(T))

constexpr uint32_t DCP_NEVER = 0x7fc00001u; /* a NaN no start state ever is: "no table yet" in cen[] */

__global__ void msd_dcp_init_kernel(DcpCtl *ctl, uint32_t *S, uint32_t *cen, uint32_t *ever, uint32_t nb, const float *state)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        DcpCtl c = {};
        c.zfront[0] = __float_as_uint(state[0]);
        c.zfront[1] = __float_as_uint(state[1]);
        *ctl = c;
    }
    if (i < nb) {
        for (int ch = 0; ch < 2; ++ch) {
            S[ch * nb + i] = __float_as_uint(state[ch]); /* first guess: the DC estimate stays where the last batch left it (it moves by
                                                            its own noise only); the first pass is then the linear prediction from there */
            cen[ch * nb + i] = DCP_NEVER;
            ever[ch * nb + i] = 0u;
        }
    }
}

constexpr int DCP_GROUP = 256; /* samples converted at a time: four per lane, their terms f * dc_a through the LDS */

/* the four samples 4 * lane ... + 3 of the group at sample g0 (a multiple of 64), channel ch, as terms f * dc_a */
template <int FMT>
__device__ __forceinline__ void dcp_terms(const uint8_t *iq, uint64_t g0, uint64_t nsamples, int lane, int ch, float dc_a, float out[4])
{
    const uint64_t g = g0 + 4u * (uint32_t)lane;
    if (g + 4 <= nsamples) {
        if (FMT == MSD_FMT_UC8) {
            const uint2 v = *reinterpret_cast<const uint2 *>(iq + 2 * g); /* I0 Q0 I1 Q1 | I2 Q2 I3 Q3 */
            const uint32_t lo = v.x >> (8 * ch), hi = v.y >> (8 * ch);
            out[0] = (((float)(lo & 0xffu) - 127.5f) / 127.5f) * dc_a; /* convert.c:133-134: a real division */
            out[1] = (((float)((lo >> 16) & 0xffu) - 127.5f) / 127.5f) * dc_a;
            out[2] = (((float)(hi & 0xffu) - 127.5f) / 127.5f) * dc_a;
            out[3] = (((float)((hi >> 16) & 0xffu) - 127.5f) / 127.5f) * dc_a;
        } else {
            const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f); /* exact: powers of two */
            const uint4 v = *reinterpret_cast<const uint4 *>(iq + 4 * g);
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[q] = ((float)(int)(int16_t)((wv[q] >> (16 * ch)) & 0xffffu) * inv) * dc_a;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[q] = g + q < nsamples ? dcp_sample<FMT>(iq, g + q, ch) * dc_a : 0.0f;
    }
}

/* One wavefront: block i of channel ch from 64 candidate start states around sbits; its prepared table into E, the states
 * in front of every 64th sample on the centre candidate's chain into fine.  tbuf: 2 x DCP_GROUP floats of LDS of its own. */
template <int FMT>
__device__ __forceinline__ void dcp_eval_block(const uint8_t *__restrict__ iq, uint64_t nsamples, uint32_t L, float dc_a, float dc_b,
                                               uint32_t sbits, uint32_t row, uint32_t i, int ch, float4 *__restrict__ E,
                                               uint32_t *__restrict__ fine, uint64_t nfine, float (*tbuf)[DCP_GROUP], int lane)
{
    const uint64_t base = (uint64_t)i * L;
    const uint32_t cnt = nsamples - base < (uint64_t)L ? (uint32_t)(nsamples - base) : L;
    const float zstart = __uint_as_float(dcp_unord(dcp_ord(sbits) + dcp_offset(lane)));
    float z = zstart;
    const uint32_t ngroups = (cnt + DCP_GROUP - 1) / DCP_GROUP;
    uint32_t *fine_row = fine + (uint64_t)ch * nfine + base / DCP_FINE;
    float cur[4], nxt[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    dcp_terms<FMT>(iq, base, nsamples, lane, ch, dc_a, cur);
    for (uint32_t g = 0; g < ngroups; ++g) {
        float *tb = tbuf[g & 1u];
        *reinterpret_cast<float4 *>(tb + 4 * lane) = make_float4(cur[0], cur[1], cur[2], cur[3]);
        if (g + 1 < ngroups) /* the next group's samples, under this group's chain (1 us of it: an HBM round trip fits) */
            dcp_terms<FMT>(iq, base + (uint64_t)(g + 1) * DCP_GROUP, nsamples, lane, ch, dc_a, nxt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); /* LDS accesses of one wavefront are served in order: the terms are there */
        const uint32_t steps = cnt - g * DCP_GROUP < (uint32_t)DCP_GROUP ? cnt - g * DCP_GROUP : (uint32_t)DCP_GROUP;
        const uint32_t quads = steps / 4u, chunks = quads / 16u; /* chunks of 64 samples: one fine state each */
        const float4 *tq = reinterpret_cast<const float4 *>(tb); /* wave-uniform addresses below: broadcasts */
        float4 a[4] = {tq[0], tq[1], tq[2], tq[3]};
        for (uint32_t c = 0; c < chunks; ++c) {
            if (lane == 31)
                fine_row[g * (DCP_GROUP / DCP_FINE) + c] = __float_as_uint(z); /* the state in front of every 64th sample */
#pragma unroll
            for (int h = 0; h < 4; ++h) { /* sixteen steps on the terms read a round earlier, the next sixteen on their way */
                const uint32_t nq = (16u * c + 4u * (uint32_t)h + 4u) & (DCP_GROUP / 4 - 1); /* (wraps at the end: read, not used) */
                const float4 n0 = tq[nq], n1 = tq[nq + 1], n2 = tq[nq + 2], n3 = tq[nq + 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    DCP_STEP(z, a[q].x, dc_b);
                    DCP_STEP(z, a[q].y, dc_b);
                    DCP_STEP(z, a[q].z, dc_b);
                    DCP_STEP(z, a[q].w, dc_b);
                }
                a[0] = n0, a[1] = n1, a[2] = n2, a[3] = n3;
            }
        }
        if (16u * chunks < quads && lane == 31)
            fine_row[g * (DCP_GROUP / DCP_FINE) + chunks] = __float_as_uint(z);
        for (uint32_t q = 16u * chunks; q < quads; ++q) { /* the ragged end of the batch */
            const float4 t4 = tq[q];
            DCP_STEP(z, t4.x, dc_b);
            DCP_STEP(z, t4.y, dc_b);
            DCP_STEP(z, t4.z, dc_b);
            DCP_STEP(z, t4.w, dc_b);
        }
        if (quads == 16u * chunks && 4u * quads < steps && lane == 31)
            fine_row[g * (DCP_GROUP / DCP_FINE) + chunks] = __float_as_uint(z);
        for (uint32_t j = 4u * quads; j < steps; ++j) { /* the ragged end of the batch; a multiple of 64 came before it */
            const float t = tb[j];
            DCP_STEP(z, t, dc_b);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            cur[q] = nxt[q];
        __builtin_amdgcn_wave_barrier(); /* the buffer written two groups from now is this group's: every lane is done with it */
    }
    /* The block's table, prepared for the walk: lane j owns the bracket [c_j, c_(j+1)) -- its ends, the table value at its
     * lower end and the secant across it.  A slope of -0.0 marks a bracket across which the table is flat: the block's map
     * is monotone, so every state inside such a bracket is mapped to that very value, exactly.
     * A state the candidates do not reach (they span 2^30 units in the last place either side, which is not the way from a
     * guess of one sign to a truth of the other) is extrapolated from the end candidate with the secant between that end and
     * the centre -- at that scale the map is affine, while the end bracket itself may be flat in float: lane 63's slope for
     * the top, and for the bottom a second slope that travels in lane 63's unused upper-end field. */
    const float x0 = zstart, y0 = z;
    float x_up = __shfl_down(zstart, 1);
    const float y_up = __shfl_down(z, 1);
    const float dx = x_up - x0, dy = y_up - y0;
    float slope = (lane < 63 && dx > 0.0f && dy > 0.0f) ? dy * __builtin_amdgcn_rcpf(dx) : 0.0f; /* (a guess's slope: any value will do) */
    if (lane < 63 && __float_as_uint(y_up) == __float_as_uint(y0))
        slope = -0.0f;
    const float xc = __shfl(zstart, 31), yc = __shfl(z, 31);
    const float x_lo = __shfl(zstart, 0), y_lo = __shfl(z, 0);
    if (lane == 63) {
        const float ex = x0 - xc, ey = y0 - yc, fx = xc - x_lo, fy = yc - y_lo;
        slope = (ex > 0.0f && ey > 0.0f) ? ey * __builtin_amdgcn_rcpf(ex) : 1.0f;
        x_up = (fx > 0.0f && fy > 0.0f) ? fy * __builtin_amdgcn_rcpf(fx) : 1.0f; /* the slope below c_0 */
    }
    E[(uint64_t)row * 64u + lane] = make_float4(x0, x_up, y0, slope);
}

template <int FMT>
__global__ void __launch_bounds__(256) msd_dcp_eval_kernel(const uint8_t *__restrict__ iq, uint64_t nsamples, uint32_t L,
                                                           float dc_a, float dc_b, const uint32_t *__restrict__ S,
                                                           uint32_t *__restrict__ cen, float4 *__restrict__ E,
                                                           uint32_t *__restrict__ fine, uint32_t nb, uint64_t nfine)
{
    __shared__ __attribute__((aligned(16))) float tbuf[4][2][DCP_GROUP]; /* [wavefront][group parity][sample of the group] */
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4u + (uint32_t)wv;
    const uint32_t i = w >> 1;
    const int ch = (int)(w & 1u);
    if (i >= nb)
        return; /* the wavefronts of a workgroup never meet: no barrier below */
    const uint32_t row = (uint32_t)ch * nb + i;
    const uint32_t sbits = S[row];
    if (cen[row] == sbits)
        return; /* evaluated around this very start state already */
    dcp_eval_block<FMT>(iq, nsamples, L, dc_a, dc_b, sbits, row, i, ch, E, fine, nfine, tbuf[wv], lane);
    if (lane == 0)
        cen[row] = sbits;
}

/* bounded waits of the fused kernel */
constexpr uint32_t DCP_SPINS = 1u << 19; /* x (a load or two + s_sleep 16, about 1.4 us): most of a second */
__device__ __forceinline__ uint32_t dcp_ld_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t dcp_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dcp_st_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

/* One walk of channel ch by the calling workgroup (256 threads: the first wavefront walks, the others fetch the next tile of
 * tables into LDS), from the frontier in ctl.  S[i] becomes the start state the walk arrived at (exact or guessed).  A lone
 * wavefront issues an instruction every five to eight cycles, so what a block costs the walk is its instruction count: the
 * tables come prepared (dcp_eval_block's last lines), one ds_read_b128 per lane and block, and the order of the floats is
 * the order of the states (-0 = +0), so nothing is converted.
 * FUSED: the tables of pass `pass` are waited for block by block (ever[row] > pass), progress is published tile by tile.
 * Returns (to every thread) false when a wait ran out. */
template <int TILE, bool FUSED>
__device__ __forceinline__ bool dcp_walk(DcpCtl *ctl, uint32_t *S, const float4 *E, const uint32_t *ever, uint32_t nb, int ch, uint32_t pass,
                                         float4 (*et)[TILE * 64], uint32_t *sh_fail)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = ctl->frontier[ch];
    uint32_t zb = ctl->zfront[ch]; /* wave-uniform */
    const uint32_t ntiles = (nb - i0 + TILE - 1) / TILE;
    if (tid == 0)
        *sh_fail = 0u;
    if (FUSED && tid == 0)
        dcp_st_rel(&ctl->prog[ch], pass * (nb + 1u) + i0); /* the blocks in front of the frontier are final */
    __syncthreads();
    auto fetch = [&](uint32_t t, int first, int nthr) { /* tile t into its buffer, by threads first ... first + nthr - 1 */
        const uint32_t r0 = i0 + t * TILE, rows = nb - r0 < (uint32_t)TILE ? nb - r0 : (uint32_t)TILE;
        if (FUSED) { /* every fetching wavefront waits for all the tile's tables itself (lanes 0 ... rows - 1) */
            bool ready = false;
            for (uint32_t spin = 0; spin < DCP_SPINS; ++spin) {
                const bool ok = (uint32_t)lane >= rows || dcp_ld(&ever[(uint32_t)ch * nb + r0 + (uint32_t)(lane % TILE)]) > pass;
                if (__ballot(ok) == ~0ull) {
                    ready = true;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* polled with plain coherent loads: one acquire at the end */
                    break;
                }
                if (dcp_ld(&ctl->gaveup))
                    break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (!ready) {
                if (lane == 0)
                    *sh_fail = 1u;
                return;
            }
        }
        const float4 *src = E + ((uint64_t)ch * nb + r0) * 64u;
        float4 *dst = et[t & 1u];
        for (uint32_t k = (uint32_t)(tid - first); k < rows * 64u; k += (uint32_t)nthr)
            dst[k] = src[k];
    };
    fetch(0, 0, 256);
    __syncthreads();
    bool exact = true;
    uint32_t frontier = nb, zfront = 0, guessed = 0;
    for (uint32_t t = 0; t < ntiles && !*sh_fail; ++t) {
        if (wave > 0) {
            if (t + 1 < ntiles)
                fetch(t + 1, 64, 192);
        } else {
            const uint32_t r0 = i0 + t * TILE, rows = nb - r0 < (uint32_t)TILE ? nb - r0 : (uint32_t)TILE;
            const float4 *eb = et[t & 1u];
            float4 nx = eb[lane];
            for (uint32_t r = 0; r < rows; ++r) {
                const uint32_t row = (uint32_t)ch * nb + r0 + r;
                const float4 p = nx; /* x: the bracket's lower end, y: its upper end, z: the table at the lower end, w: the secant */
                /* the table's centre is the start state it was evaluated around (S[row], but for the sign of a zero) */
                const uint32_t sold = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.x), 31);
                if (r + 1 < rows)
                    nx = eb[(r + 1) * 64u + (uint32_t)lane]; /* does not depend on this block's outcome */
                const float zf = __uint_as_float(zb);
                /* The candidates ascend with the lanes, so the lanes whose bracket starts at or below Z are a prefix: its last
                 * lane owns Z (lane 63 also what lies beyond c_63; below c_0 nobody does: lane 0 extrapolates with the slope
                 * that travels in lane 63's upper-end field). */
                const uint64_t m_above = __ballot(p.x <= zf);
                const uint64_t m_sure = __ballot(p.x == zf) | __ballot(__float_as_uint(p.w) == 0x80000000u); /* Z is the candidate, or the bracket is flat */
                const float guess = (zf - p.x) * p.w + p.z; /* = the table's value itself where the lane is sure */
                uint32_t nz;
                bool ex;
                if (m_above) {
                    const int owner = __popcll(m_above) - 1;
                    nz = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(guess), owner);
                    ex = ((m_sure >> owner) & 1ull) != 0;
                } else {
                    const float s_low = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.y), 63));
                    const float x_lo = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.x), 0));
                    const float y_lo = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.z), 0));
                    nz = __float_as_uint((zf - x_lo) * s_low + y_lo);
                    ex = false;
                }
                if (!ex)
                    ++guessed;
                if (exact && !ex) { /* this block's start is exact, its end is not: the next pass starts here */
                    frontier = r0 + r;
                    zfront = zb;
                    exact = false;
                }
                if (lane == 0 && zb != sold)
                    S[row] = zb;
                zb = nz;
            }
            if (FUSED && ((t & 3u) == 3u || t + 1 == ntiles)) { /* the start states up to here are written: the evaluation may go on with
                                                                   these blocks (every fourth tile: a release is microseconds) */
                __threadfence();
                if (lane == 0)
                    dcp_st_rel(&ctl->prog[ch], pass * (nb + 1u) + r0 + rows);
            }
        }
        __syncthreads();
    }
    const bool failed = *sh_fail != 0u;
    if (tid == 0 && !failed) {
        ctl->passes_ch[ch] += 1u;
        atomicAdd(&ctl->guessed, guessed);
        if (exact) {
            ctl->zend[ch] = zb;
            ctl->frontier[ch] = nb;
            __threadfence();
            dcp_st_rel(&ctl->done_ch[ch], 1u);
            if (atomicAdd(&ctl->ndone, 1u) == 1u)
                dcp_st_rel(&ctl->done, 1u); /* the other channel got there before */
        } else {
            ctl->zfront[ch] = zfront;
            __threadfence();
            dcp_st_rel(&ctl->frontier[ch], frontier);
        }
    }
    __syncthreads();
    return !failed;
}

__global__ void __launch_bounds__(256) msd_dcp_walk_kernel(DcpCtl *ctl, uint32_t *S, const float4 *__restrict__ E, uint32_t nb)
{
    __shared__ float4 et[2][64 * 64];
    __shared__ uint32_t sh_fail;
    const int ch = blockIdx.x;
    if (ctl->done || ctl->done_ch[ch])
        return;
    dcp_walk<64, false>(ctl, S, E, nullptr, nb, ch, 0u, et, &sh_fail);
}

/* Behind the passes: where the in-order kernel has to take over, if at all. */
__global__ void msd_dcp_handover_kernel(DcpCtl *ctl, const uint32_t *S, uint32_t nb, uint32_t L, uint64_t nsamples)
{
    if (threadIdx.x || blockIdx.x)
        return;
    const uint32_t f0 = ctl->done_ch[0] ? nb : ctl->frontier[0], f1 = ctl->done_ch[1] ? nb : ctl->frontier[1];
    const uint32_t f = f0 < f1 ? f0 : f1; /* the starts of blocks 0 ... f are exact in both channels */
    uint64_t resume = (uint64_t)f * L;
    if (resume > nsamples || ctl->done)
        resume = nsamples;
    ctl->resume_lo = (uint32_t)resume;
    ctl->resume_hi = (uint32_t)(resume >> 32);
    ctl->resume_z[0] = f < nb ? S[f] : ctl->zend[0];
    ctl->resume_z[1] = f < nb ? S[nb + f] : ctl->zend[1];
}

/* All passes in one cooperative launch: workgroups 0 and 1 walk channels 0 and 1, every wavefront of the others owns one
 * (block, channel).  See the head of the file. */
constexpr int DCP_FTILE = 16;
template <int FMT>
__global__ void __launch_bounds__(256) msd_dcp_fused_kernel(const uint8_t *__restrict__ iq, uint64_t nsamples, uint32_t L, float dc_a, float dc_b,
                                                            DcpCtl *ctl, uint32_t *S, uint32_t *cen, float4 *E, uint32_t *fine, uint32_t *ever,
                                                            uint32_t nb, uint64_t nfine, uint32_t max_passes)
{
    __shared__ float4 et[2][DCP_FTILE * 64]; /* the walk's two tiles; an evaluating workgroup's terms (8 KB) lie in the same bytes */
    __shared__ uint32_t sh_fail;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x < 2u) {
        const int ch = (int)blockIdx.x;
        bool ok = true;
        uint32_t w = 0;
        for (; w < max_passes && ok; ++w) {
            ok = dcp_walk<DCP_FTILE, true>(ctl, S, E, ever, nb, ch, w, et, &sh_fail);
            if (dcp_ld_acq(&ctl->done_ch[ch]) || dcp_ld(&ctl->gaveup))
                break;
        }
        if (threadIdx.x == 0 && !dcp_ld(&ctl->done_ch[ch]))
            dcp_st_rel(&ctl->gaveup, 1u); /* the passes (or a wait) ran out: the in-order kernel takes the batch */
        return;
    }
    float(*tbuf)[DCP_GROUP] = reinterpret_cast<float(*)[DCP_GROUP]>(reinterpret_cast<float *>(&et[0][0]) + (size_t)wv * 2 * DCP_GROUP);
    const uint32_t w = (blockIdx.x - 2u) * 4u + (uint32_t)wv;
    const uint32_t i = w >> 1;
    const int ch = (int)(w & 1u);
    if (i >= nb)
        return;
    const uint32_t row = (uint32_t)ch * nb + i;
    for (uint32_t p = 0; p <= max_passes; ++p) {
        bool stop = false;
        if (p > 0) { /* walk p - 1 has gone past this block (or never will: it lies in front of the frontier) */
            const uint32_t want = (p - 1u) * (nb + 1u) + i + 1u;
            bool there = false;
            for (uint32_t spin = 0; spin < DCP_SPINS; ++spin) {
                there = dcp_ld(&ctl->prog[ch]) >= want;
                stop = dcp_ld(&ctl->done_ch[ch]) || dcp_ld(&ctl->gaveup);
                if (there || stop) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* polled with plain coherent loads: one acquire at the end */
                    break;
                }
                __builtin_amdgcn_s_sleep(127); /* 3 us: a thousand wavefronts ask the same word */
            }
            if (!there && !stop) {
                if (lane == 0)
                    dcp_st_rel(&ctl->gaveup, 1u);
                return;
            }
        }
        const uint32_t sbits = dcp_ld(&S[row]);
        if (dcp_ld(&cen[row]) != sbits) {
            dcp_eval_block<FMT>(iq, nsamples, L, dc_a, dc_b, sbits, row, i, ch, E, fine, nfine, tbuf, lane);
            if (lane == 0)
                cen[row] = sbits;
        }
        __threadfence(); /* every lane's part of the table, then the counter */
        if (lane == 0)
            dcp_st_rel(&ever[row], p + 1u);
        if (stop || dcp_ld_acq(&ctl->frontier[ch]) > i)
            return; /* the channel is done, or this block's start is final: nobody will ask for its table again */
    }
}

/* One lane per 64 samples, from the exact state in front of them: what msd_dcfilter_kernel's workers compute (convert.c:133-151). */
template <int FMT>
__global__ void __launch_bounds__(256) msd_dcp_out_kernel(const uint8_t *__restrict__ iq, uint64_t nsamples, float dc_a, float dc_b,
                                                          const uint32_t *__restrict__ fine, uint64_t nfine, const DcpCtl *ctl,
                                                          float *state, uint16_t *__restrict__ mag, float *__restrict__ magsq_out)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && ctl->done) { /* (a batch the passes did not finish: the in-order kernel behind this one leaves the state) */
        state[0] = __uint_as_float(ctl->zend[0]);
        state[1] = __uint_as_float(ctl->zend[1]);
    }
    const uint64_t base = g * DCP_FINE;
    const uint64_t limit = (uint64_t)ctl->resume_lo | ((uint64_t)ctl->resume_hi << 32); /* nsamples when the batch is exact */
    if (base >= nsamples || base >= limit)
        return; /* (the blocks from `limit` on are the in-order kernel's) */
    float zi = __uint_as_float(fine[g]), zq = __uint_as_float(fine[nfine + g]);
    const uint32_t cnt = nsamples - base < (uint64_t)DCP_FINE ? (uint32_t)(nsamples - base) : (uint32_t)DCP_FINE;
    for (uint32_t j0 = 0; j0 < cnt; j0 += 8) {
        const uint32_t m = cnt - j0 < 8u ? cnt - j0 : 8u;
        float fi[8], fq[8];
        if (m == 8u) { /* 8 samples = one 16-byte (UC8) or two 16-byte loads; base is a multiple of 64 samples */
            if (FMT == MSD_FMT_UC8) {
                const uint4 v = *reinterpret_cast<const uint4 *>(iq + 2 * (base + j0));
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fi[2 * q] = ((float)(wv[q] & 0xffu) - 127.5f) / 127.5f;
                    fq[2 * q] = ((float)((wv[q] >> 8) & 0xffu) - 127.5f) / 127.5f;
                    fi[2 * q + 1] = ((float)((wv[q] >> 16) & 0xffu) - 127.5f) / 127.5f;
                    fq[2 * q + 1] = ((float)(wv[q] >> 24) - 127.5f) / 127.5f;
                }
            } else {
                const float inv = (FMT == MSD_FMT_SC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
                const uint4 v0 = *reinterpret_cast<const uint4 *>(iq + 4 * (base + j0));
                const uint4 v1 = *reinterpret_cast<const uint4 *>(iq + 4 * (base + j0) + 16);
                const uint32_t wv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    fi[q] = (float)(int)(int16_t)(wv[q] & 0xffffu) * inv;
                    fq[q] = (float)(int)(int16_t)(wv[q] >> 16) * inv;
                }
            }
        } else {
            for (uint32_t q = 0; q < 8u; ++q) {
                fi[q] = fq[q] = 0.0f;
                if (q < m) {
                    fi[q] = dcp_sample<FMT>(iq, base + j0 + q, 0);
                    fq[q] = dcp_sample<FMT>(iq, base + j0 + q, 1);
                }
            }
        }
        uint32_t mg[8];
        float sq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float ti = fi[q] * dc_a, tq = fq[q] * dc_a;
            const float pi = zi * dc_b, pq = zq * dc_b;
            zi = ti + pi; /* -ffp-contract=off: separately rounded, convert.c:137-138 */
            zq = tq + pq;
            const float di = fi[q] - zi, dq = fq[q] - zq;
            const float sq_i = di * di, sq_q = dq * dq;
            float magsq = sq_i + sq_q;
            magsq = fminf(magsq, 1.0f); /* convert.c:144-145 */
            const float mm = __builtin_sqrtf(magsq);
            mg[q] = (uint32_t)(uint16_t)(mm * 65535.0f + 0.5f);
            sq[q] = magsq;
        }
        if (m == 8u) {
            uint4 o;
            o.x = mg[0] | (mg[1] << 16), o.y = mg[2] | (mg[3] << 16), o.z = mg[4] | (mg[5] << 16), o.w = mg[6] | (mg[7] << 16);
            *reinterpret_cast<uint4 *>(mag + base + j0) = o;
            *reinterpret_cast<float4 *>(magsq_out + base + j0) = make_float4(sq[0], sq[1], sq[2], sq[3]);
            *reinterpret_cast<float4 *>(magsq_out + base + j0 + 4) = make_float4(sq[4], sq[5], sq[6], sq[7]);
        } else {
            for (uint32_t q = 0; q < m; ++q) {
                mag[base + j0 + q] = (uint16_t)mg[q];
                magsq_out[base + j0 + q] = sq[q];
            }
        }
    }
}

uint32_t dcp_blocks(uint64_t nsamples, uint32_t L) { return (uint32_t)((nsamples + L - 1) / L); }

} // namespace

/* The workspace of one batch of at most max_samples in blocks of block_len samples (0: of msd_dcp_block_len(n) for any n
 * up to max_samples: at most 1025 blocks below 32 Mi samples, blocks of 32768 beyond). */
static size_t dcp_e_offset(uint64_t nb) { return 256 + (((size_t)nb * 2 * 4 * 3 + 255) & ~(size_t)255); } /* ctl; S, cen, ever */
extern "C" size_t msd_dcp_work_bytes(uint64_t max_samples, uint32_t block_len)
{
    uint64_t nb = block_len ? dcp_blocks(max_samples, block_len) + 1 : max_samples / 32768u + 2;
    if (!block_len && nb < 1032)
        nb = 1032;
    const uint64_t nfine = (max_samples + DCP_FINE - 1) / DCP_FINE + 1;
    return dcp_e_offset(nb) + (size_t)nb * 2 * 64 * 16 + (size_t)nfine * 2 * 4 + 512;
}

/* Block length for a batch: a power of two between 1024 and 65536 samples that leaves about a thousand blocks -- the
 * evaluation is one wavefront per block and channel, latency-bound at 8 ns per sample of a block with one or two wavefronts
 * per SIMD and issue-bound at 3 ns per sample and SIMD beyond four (scripts/micro/dcp_chain_occupancy.hip); the walk costs
 * 0.17 us per block.  One buffer of 131072 samples is best served by 64 blocks of 2048 (16 + 11 us per pass); beyond 64 Mi
 * samples the evaluation is issue-bound whatever the block length and the walk is what a longer block saves: 65536. */
extern "C" uint32_t msd_dcp_block_len(uint64_t nsamples)
{
    uint32_t L = nsamples >= 65536u ? 2048 : 1024;
    while (L < 32768u && nsamples / L > 1024u)
        L *= 2;
    if (nsamples / L > 2048u)
        L *= 2;
    return L;
}

/* How many workgroups of the fused kernel can be resident at once on the current device (0: no cooperative launch). */
template <int FMT>
static int dcp_fused_capacity()
{
    static int cap[16] = {0}; /* per device, filled in once: -1 none */
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
        return 0;
    if (cap[dev] == 0) {
        int coop = 0, per_cu = 0, cus = 0;
        cap[dev] = -1;
        if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess && coop &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(msd_dcp_fused_kernel<FMT>), 256, 0) == hipSuccess &&
            per_cu > 0 && cus > 0)
            cap[dev] = per_cu * cus;
    }
    return cap[dev] > 0 ? cap[dev] : 0;
}

template <int FMT>
static void dcp_launch(const uint8_t *iq, uint64_t n, uint32_t L, float dc_a, float dc_b, float *d_state, uint16_t *d_mag,
                       float *d_magsq, uint8_t *work, int max_passes, int fused, hipStream_t stream)
{
    uint32_t nb = dcp_blocks(n, L);
    uint64_t nfine = (n + DCP_FINE - 1) / DCP_FINE;
    DcpCtl *ctl = reinterpret_cast<DcpCtl *>(work);
    uint32_t *S = reinterpret_cast<uint32_t *>(work + 256);
    uint32_t *cen = S + 2 * (size_t)nb;
    uint32_t *ever = cen + 2 * (size_t)nb;
    float4 *E = reinterpret_cast<float4 *>(work + dcp_e_offset(nb));
    uint32_t *fine = reinterpret_cast<uint32_t *>(E + (size_t)nb * 2 * 64);
    hipLaunchKernelGGL(msd_dcp_init_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, ctl, S, cen, ever, nb, d_state);
    const uint32_t eval_grid = (2 * nb + 3) / 4;
    bool done_fused = false;
    if (fused && (int)(eval_grid + 2u) <= dcp_fused_capacity<FMT>()) {
        uint32_t mp = (uint32_t)max_passes;
        void *args[] = {&iq, &n, &L, &dc_a, &dc_b, &ctl, &S, &cen, &E, &fine, &ever, &nb, &nfine, &mp};
        done_fused = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(msd_dcp_fused_kernel<FMT>), dim3(eval_grid + 2u), dim3(256), args, 0,
                                                stream) == hipSuccess;
        if (!done_fused)
            (void)hipGetLastError();
    }
    if (!done_fused) {
        for (int p = 0; p < max_passes; ++p) {
            hipLaunchKernelGGL(msd_dcp_eval_kernel<FMT>, dim3(eval_grid), dim3(256), 0, stream, iq, n, L, dc_a, dc_b, S, cen, E, fine, nb, nfine);
            hipLaunchKernelGGL(msd_dcp_walk_kernel, dim3(2), dim3(256), 0, stream, ctl, S, E, nb);
        }
    }
    /* the blocks whose table is not centred on their final start state: the states at every 64th sample */
    hipLaunchKernelGGL(msd_dcp_eval_kernel<FMT>, dim3(eval_grid), dim3(256), 0, stream, iq, n, L, dc_a, dc_b, S, cen, E, fine, nb, nfine);
    hipLaunchKernelGGL(msd_dcp_handover_kernel, dim3(1), dim3(64), 0, stream, ctl, S, nb, L, n);
    hipLaunchKernelGGL(msd_dcp_out_kernel<FMT>, dim3((unsigned)((nfine + 255) / 256)), dim3(256), 0, stream, iq, n, dc_a, dc_b, fine, nfine, ctl,
                       d_state, d_mag, d_magsq);
}

/* d_work: msd_dcp_work_bytes(>= nsamples, block_len) of device memory; behind this call the caller queues
 * msd_launch_dcfilter(..., skip_if = d_work), which does the batch in order if the passes did not get there. */
extern "C" int msd_launch_dcfilter_parallel(int format, const void *d_iq, uint64_t nsamples, float dc_a, float dc_b, float *d_state,
                                            uint16_t *d_mag, float *d_magsq, void *d_work, uint32_t block_len, int max_passes, int fused,
                                            hipStream_t stream)
{
    const uint8_t *iq = static_cast<const uint8_t *>(d_iq);
    uint8_t *work = static_cast<uint8_t *>(d_work);
    if (nsamples == 0)
        return 0;
    if (!d_work || block_len < 64u || (block_len % DCP_FINE) != 0u || max_passes < 1 || (reinterpret_cast<uintptr_t>(d_iq) & 15u))
        return -22;
    switch (format) {
    case MSD_FMT_UC8:
        dcp_launch<MSD_FMT_UC8>(iq, nsamples, block_len, dc_a, dc_b, d_state, d_mag, d_magsq, work, max_passes, fused, stream);
        break;
    case MSD_FMT_SC16:
        dcp_launch<MSD_FMT_SC16>(iq, nsamples, block_len, dc_a, dc_b, d_state, d_mag, d_magsq, work, max_passes, fused, stream);
        break;
    case MSD_FMT_SC16Q11:
        dcp_launch<MSD_FMT_SC16Q11>(iq, nsamples, block_len, dc_a, dc_b, d_state, d_mag, d_magsq, work, max_passes, fused, stream);
        break;
    default:
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
