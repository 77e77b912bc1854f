/* msd_kernels.h -- launch interface between msd_capi.cpp and msd_kernels.hip */
#ifndef MSD_KERNELS_H
#define MSD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "modes_hip.h"
#include "msd_internal.h"


/* a record as the emit code writes it: the message, nothing else (the power sum travels in signalLevel) */
typedef struct msd_wire {
    msd_message mm;
} msd_wire;

/* The records of one batch (Mode S messages, then Mode A/C replies, per buffer) for the wavefronts of the NEXT
 * batch's scan kernel to write on their way in: wavefront w * stride takes buffer w.  nbuffers == 0: nothing. */
typedef struct MsdEmitJob {
    uint32_t nbuffers, stride, cap;
    const uint64_t *totals;    /* [2]: arena overflow flag of that batch */
    const uint32_t *nmsgs;     /* [buffer] */
    const uint32_t *rec_off;   /* [buffer] records in front of the buffer's (msd_power_buffers_kernel), or NULL: the
                                  wavefronts add up nmsgs / nac themselves */
    const struct msd_acc *acc; /* [buffer][MSD_RB_MSG_CAP] */
    const msd_try *tries;      /* that batch's dense try list */
    const uint64_t *ts;        /* [buffer][2] */
    const unsigned long long *power; /* [buffer][MSD_RB_MSG_CAP] signal power sums */
    msd_wire *dense;           /* out: cap records, pinned host memory */
    unsigned long long *side;  /* out: per record, power sum | signal_len << 48 (0: Mode A/C), pinned host memory */
    const msd_ac_hit *ac;      /* NULL: Mode A/C off */
    const uint64_t *ac_totals;
    const uint32_t *acc_ac;
    const uint32_t *nac;
} MsdEmitJob;

typedef struct MsdScanParams {
    const uint8_t *iq;        /* first sample of this batch (16-byte aligned) */
    const uint8_t *prev_tail; /* the MSD_HALO_FRONT samples before it */
    const uint8_t *ragged;    /* 32 readable bytes holding the last nsamples % 8 samples, zero padded */
    uint16_t *mag_out;        /* or NULL: the batch's magnitudes, sample by sample (zero where there is none), for the Mode A/C
                                 candidate kernel behind the scan -- which then converts nothing a second time */
    int have_prev;            /* 0: start of stream or discontinuity -> zero magnitudes (fifo.c:180) */
    int threshold;            /* Modes.preambleThreshold */
    uint64_t batch_first;     /* absolute index of iq[0]; multiple of MSD_CHUNK_SAMPLES */
    uint64_t nsamples;
    uint32_t ntiles;       /* tiles of msd_scan_tile(format) scan positions */
    uint32_t tiles_per_wg; /* tiles per region (= per wavefront of the scan kernel) */
    const uint16_t *lut;
    const uint32_t *crc_tab;
    const uint32_t *syn56;
    const uint32_t *syn112;
    uint32_t nsyn56, nsyn112;
    const uint32_t *synhash; /* msd_tables.synhash (MSD_SYNH_WORDS dwords) and its two multipliers */
    uint32_t synh_mul56, synh_mul112;
    const uint32_t *slicer; /* msd_tables.slicer: MSD_SLICER_WORDS dwords */
    const uint64_t *fix2_56, *fix2_112; /* --aggressive: the two-bit correction hash tables (msd_fix2_table);
                                           NULL otherwise */
    uint32_t fix2_lg56, fix2_lg112;
    msd_hit *hits;
    msd_try *tries;
    uint32_t hcap, tcap; /* per-region capacities */
    msd_region_counts *counts; /* [scan workgroups * MSD_SCAN_WAVES] */
    msd_wg_totals *wg_totals;  /* [scan workgroups] */
    uint64_t *chunk_sums; /* [buffers in batch][2]: sum of mag, sum of mag^2 */
    float *tile_sums;     /* SC16 / SC16Q11: [tile][2] approximate float sums of a tile's magnitudes and squares (what the
                             float-sum kernel predicts its binades from), or NULL */
    /* ---- lean layout (msd_capi.cpp): no gather kernel behind the scan ----
     * regions_per_buffer != 0: region r is piece r % k of buffer r / k (tiles_per_region tiles, never across a buffer
     * boundary), so that the resolve workgroup of a buffer reads its k region slices where they are; the try index
     * in a hit record is then arena-absolute (region * tcap + index). */
    uint32_t regions_per_buffer, tiles_per_region;
    unsigned long long *overflow; /* the batch's totals[2]: set by any region whose slice overflowed */
    /* predicted adds, written as the tries are found: every CRC-clean DF17 / DF11(II=0) address with the first buffer
     * holding one (msd_internal.h); NULL: nobody wants them */
    unsigned long long *pred;
    uint32_t pred_gen;
    /* the batch's last MSD_HALO_FRONT samples, kept for the look-behind of its successor (lean layout: the first
     * wavefront copies them on its way in; otherwise the gather kernel does) */
    const uint32_t *tail_src;
    uint32_t *tail_dst;
    uint32_t tail_words;
    unsigned long long *timers; /* MSD_KERNEL_TIMING builds only */
    int debug_flags;      /* MSD_DEBUG_FLAGS env, perf experiments only: 1 = stop after the scan,
                             2 = stop after the conversion (results are then incomplete) */
    MsdEmitJob emit;      /* the previous batch's records, or nbuffers == 0 */
} MsdScanParams;

typedef struct MsdResolveParams {
    const msd_hit *hits; /* the batch's ordered candidate lists, as the gather kernel left them */
    const msd_try *tries;
    const uint64_t *totals;   /* the batch's totals on the device: [0] hits, [2] arena overflow flag */
    const uint32_t *buf_first; /* [buffer + 1] where each buffer's hits start in `hits` (the gather kernel), or NULL */
    const uint32_t *valid;    /* [buffer] new samples */
    const uint64_t *ts;       /* [buffer][2] sampleTimestamp, sysTimestamp */
    const uint32_t *snaps;    /* [snapshot][MSD_SNAP_WORDS] */
    const uint32_t *snap_idx; /* [buffer] snapshot to resolve against */
    const uint32_t *todo;     /* [workgroup] buffer to resolve */
    msd_rbuf *rbuf;           /* [buffer], pinned host memory: written, never read, by the kernels */
    uint32_t *nmsgs;          /* [buffer] accepted messages, for the offsets of the emit kernel */
    msd_acc *acc;             /* [buffer][MSD_RB_MSG_CAP] */
    uint32_t *adds;           /* [buffer][MSD_RB_MSG_CAP]: the complete add lists (msd_rbuf holds the first ones) */
    const msd_ac_hit *ac;     /* Mode A/C candidates of the batch, ordered; NULL: Mode A/C off */
    const uint64_t *ac_totals; /* [0] their number, [2] arena overflow flag */
    uint32_t *acc_ac;         /* [buffer][MSD_RB_AC_CAP] indices of the accepted ones */
    uint32_t *nac;            /* [buffer] how many */
    /* first pass only: its first workgroup publishes the prediction list the predict kernel built */
    int first_pass;            /* publish the prediction list (and, lean layout, the sums and totals) */
    int ctl_implicit;          /* first pass: todo[i] = i, snap_idx = 0, valid / ts follow from the two values below */
    uint64_t sample_counter0, batch_samples; /* sample clock at the batch's first sample; its samples */
    msd_pred_entry *h_pred;    /* pinned host memory */
    uint32_t *h_pred_count;
    const unsigned long long *pred; /* the batch's prediction table (msd_internal.h: MSD_PRED_*), written by its scan */
    uint32_t pred_gen;
    /* lean layout: the candidate lists stay in the scan's region slices.  hits / tries above are then the arenas,
     * region r's hits are hits[r * hcap .. + region_counts[r].nhits), buffer b owns regions [b * k, (b + 1) * k);
     * the first pass also publishes what the gather kernel used to: the buffer's level / power sums and, from its
     * first workgroup, the batch's totals and overflow flag. */
    const msd_region_counts *region_counts; /* NULL: dense lists */
    const msd_wg_totals *wg_totals;
    uint32_t regions_per_buffer, hcap, nscan_wg, nregions; /* (a trailing buffer without samples has no regions) */
    uint64_t *sums, *h_sums, *h_totals;
    const float *fmeans; /* 16-bit IQ: the buffers' float sums (msd_float_means kernels), published like the integer ones */
    float *h_fmeans;
    uint64_t *h_ac_totals; /* Mode A/C: the candidate totals and overflow flag, likewise */
    /* power != NULL: every workgroup finishes with the signal power of its buffer's accepted messages
     * (demod_2400.c:386-399), power[buffer][MSD_RB_MSG_CAP], from the batch's samples -- no kernel of its own */
    unsigned long long *power;
    const uint8_t *iq, *prev_tail;
    int have_prev, format;
    uint64_t batch_first, nsamples;
    const uint16_t *lut;
} MsdResolveParams;

#ifdef __cplusplus
extern "C" {
#endif
/* h_* are device-visible pinned host addresses */
int msd_launch_publish(const uint64_t *totals, const uint64_t *ac_totals, uint64_t *sums, const float *fmeans,
                       uint32_t nbuffers, uint64_t *h_totals, uint64_t *h_ac_totals, uint64_t *h_sums, float *h_fmeans,
                       hipStream_t stream);
/* The prediction table of a batch (msd_internal.h) is filled by its scan kernel (MsdScanParams.pred); the list of
 * entries is published by the first resolve pass: h_pred[0..*h_pred_count) in pinned host memory, *h_pred_count =
 * MSD_PRED_LIST + 1 on overflow.  table[patches[i].slot].first = patches[i].first; patches is pinned host memory. */
int msd_launch_pred_patch(unsigned long long *table, const msd_pred_patch *patches, uint32_t n, hipStream_t stream);
int msd_launch_resolve(const MsdResolveParams *p, uint32_t ntodo, hipStream_t stream);
/* signal power of the accepted messages of every buffer: out[buffer][MSD_RB_MSG_CAP] (device); and, if rec_off is
 * not NULL, rec_off[buffer] = messages and (nac not NULL) Mode A/C replies of the buffers in front of it */
int msd_launch_power_buffers(const MsdScanParams *p, int format, const msd_acc *acc, const msd_try *tries,
                             const uint32_t *nmsgs, uint32_t nbuffers, const uint64_t *totals, unsigned long long *out,
                             const uint32_t *nac, uint32_t *rec_off, hipStream_t stream);
/* one accepted message as the emit kernel leaves it: the record, with the 64-bit signal power sum
 * sitting in the bytes of signalLevel until the host has turned it into the level (it needs the sum
 * itself for the power statistics, and 8 bytes less per message cross PCIe) */
/* the accepted messages as dense records, cap entries (what does not fit is dropped; the host notices
 * from the counts) */
int msd_launch_emit(const MsdResolveParams *p, uint32_t nbuffers, const unsigned long long *power, unsigned long long *side, msd_wire *dense,
                    msd_fields *fields /* NULL: no field decode */, uint32_t cap, hipStream_t stream);
/* the field decoder of the emit kernel on its own: out[i] = fields of in[i] (device pointers) */
int msd_launch_fields(const msd_message *d_in, msd_fields *d_out, uint32_t n, hipStream_t stream);
uint32_t msd_scan_tile(int format); /* scan positions per tile of the scan kernel for this sample format */
size_t msd_scan_lds_bytes(int format);
/* nregions wavefronts, MSD_SCAN_WAVES per workgroup */
int msd_launch_scan(const MsdScanParams *p, int format, uint32_t nregions, hipStream_t stream);
/* Regions -> dense ordered lists, one workgroup per region (nwg = the scan's nregions); the last
 * workgroup leaves the totals in `totals` and, if h_totals / h_sums are not NULL, writes them and the
 * per-buffer level/power sums to those pinned host addresses and zeroes the device sums for the slot's
 * next batch.  wipe[0..wipe_bytes) (a multiple of 16) is set to all-ones, tail_bytes (a multiple of
 * 4) are copied from tail_src to tail_dst on the way.  buf_first[b], b = 0 .. nbuffers: index in the dense list
 * of the first hit at or behind sample b * MSD_CHUNK_SAMPLES (a region is region_len scan positions). */
int msd_launch_gather(const msd_region_counts *counts, const msd_wg_totals *wg_totals, uint32_t nwg, uint64_t *totals,
                      const msd_hit *hits,
                      const msd_try *tries, uint32_t hcap, uint32_t tcap, msd_hit *dense_hits, uint64_t dense_hcap,
                      msd_try *dense_tries, uint64_t dense_tcap, uint64_t *sums, uint32_t nbuffers, uint64_t *h_totals,
                      uint64_t *h_sums, void *wipe, uint32_t wipe_bytes, const void *tail_src, void *tail_dst,
                      uint32_t tail_bytes, uint32_t region_len, uint32_t *buf_first, uint32_t try_abs, hipStream_t stream);
int msd_launch_power(const MsdScanParams *p, int format, const uint64_t *d_req, uint32_t nreq,
                     unsigned long long *d_out, hipStream_t stream);
/* Mode A/C candidate stage: noise levels (unless noise_ready), candidate kernel, ordered gather.
 * d_totals[0] receives the number of candidates, d_totals[2] an overflow flag.  phase 0: both kernels; 1: the candidate
 * kernel; 2: the gather of what phase 1 left in d_regions / d_counts (10 us of latency that the pipeline moves to the head of
 * the batch's resolve chain, off the scan stream). */
int msd_launch_ac(const MsdScanParams *p, int format, const uint64_t *d_sums, const float *d_fmeans,
                  uint32_t nbuffers, uint32_t *d_noise, int noise_ready, msd_ac_hit *d_regions,
                  uint64_t region_total, msd_wg_counts *d_counts, uint64_t *d_offsets, uint64_t *d_totals,
                  msd_ac_hit *d_dense, uint64_t dense_cap, uint32_t max_wg, int phase, hipStream_t stream);
/* SC16Q11 through the table of a -DSC16Q11_TABLE_BITS reference (convert.c:264-328): IQ -> u16 magnitudes; d_sums (or NULL)
 * receives the level / power sums the converter entry reports */
int msd_launch_q11_table(const void *d_iq, uint64_t nsamples, const uint16_t *d_table, int bits, uint16_t *d_mag,
                         unsigned long long *d_sums, int cu_count, hipStream_t stream);
int msd_launch_convert(int format, const void *d_iq, uint32_t nsamples, const uint16_t *d_lut,
                       uint16_t *d_mag, unsigned long long *d_sums, hipStream_t stream);
/* --dcfilter: IQ -> DC-blocked u16 magnitudes + f32 squares, the converter state (z1_I, z1_Q, device
 * memory) carried from call to call; then the sequential per-buffer float sums of those squares */
int msd_launch_dcfilter(int format, const void *d_iq, uint64_t nsamples, float dc_a, float dc_b, float *d_state,
                        uint16_t *d_mag, float *d_magsq, const void *d_skip_if, hipStream_t stream);
/* ... the same, exact and parallel in time (msd_dc_kernels.hip): blocks of block_len samples evaluated from 64 candidate start
 * states each, an in-order walk that is exact wherever a candidate or the monotonicity of the block's map decides, at most
 * max_passes passes.  d_work: msd_dcp_work_bytes(nsamples, block_len); its first word is 1 when the batch came out exact,
 * and msd_launch_dcfilter(..., d_skip_if = d_work, ...) queued behind finishes the batch in order when it did not -- from the first
 * block the passes did not get exact (words 16-19 of d_work: that sample and the two states in front of it).  d_iq 16-byte aligned.
 * fused: all passes in one cooperative launch where the batch's blocks can be resident together (else, and with 0, two launches per pass). */
int msd_launch_dcfilter_parallel(int format, const void *d_iq, uint64_t nsamples, float dc_a, float dc_b, float *d_state,
                                 uint16_t *d_mag, float *d_magsq, void *d_work, uint32_t block_len, int max_passes, int fused, hipStream_t stream);
size_t msd_dcp_work_bytes(uint64_t max_samples, uint32_t block_len);
uint32_t msd_dcp_block_len(uint64_t nsamples);

int msd_launch_dc_sums(const float *d_magsq, uint64_t nsamples, uint64_t buffer_len, uint32_t nbuffers, float *d_out,
                       void *d_work, int phase, hipStream_t stream);
/* tile_sums: the scan kernel's per-1024-sample approximate sums of the same batch (buffer_len a multiple of 1024), or NULL */
/* d_work: msd_fm_work_bytes(nbuffers) of device memory for the three kernels' hand-over (buffers of at most 131072 samples);
 * NULL or longer buffers: the one-wavefront-per-sum kernel */
size_t msd_fm_work_bytes(uint32_t nbuffers);
/* phase 0: all of it; 1: the block functions (vector-ALU work, behind the scan); 2: the apply walk (latency-bound: two
 * wavefronts per buffer), on any stream behind phase 1 -- the pipeline puts it at the head of the batch's resolve chain,
 * where it runs beside the next batch's kernels.  msd_fm_deferrable: the call has the two phases. */
int msd_fm_deferrable(const void *d_work, uint64_t buffer_len, uint32_t nbuffers);
int msd_launch_float_means(int format, const void *d_iq, uint64_t nsamples, uint64_t buffer_len,
                           uint32_t nbuffers, float *d_out, const float *tile_sums, void *d_work, int phase,
                           hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
