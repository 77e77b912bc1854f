/*
 * msd_internal.h -- records exchanged between the GPU candidate stage and the ordered host
 * resolve stage, and the host-side tables.  C and HIP both include this.
 *
 * Why two stages: which messages demodulate2400 accepts depends on two pieces
 * of sequential state -- the skip-ahead after an accepted message (demod_2400.c:416) and the ICAO
 * address filter that accepted DF11/DF17 messages write and every score reads
 * (mode_s.c:343-393,717-726; icao_filter.c).  Everything else is a pure function of the samples.
 * The GPU therefore evaluates, for every scan position, everything that does not depend on that
 * state, and emits two ordered lists:
 *   hits  -- one per scan position where at least one preamble test fired (demod_2400.c:298-330)
 *   tries -- one per (position, trial phase) whose score can exceed -2, i.e. is either >= 0 or
 *            depends on the address filter; tries whose score is -2 regardless of the filter
 *            (unknown DF, all-zero, uncorrectable syndrome) are only counted in the hit's mask.
 * The resolve stage replays the reference's state machine over those lists.
 */
#ifndef MSD_INTERNAL_H
#define MSD_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* geometry of the scan kernel (overridable at build time for experiments) */
#ifndef MSD_SCAN_WAVES
#define MSD_SCAN_WAVES 16 /* wavefronts per workgroup; one workgroup per CU (they share the UC8 table in LDS) */
#endif
#ifndef MSD_SCAN_WGS_PER_CU
#define MSD_SCAN_WGS_PER_CU 1 /* workgroups the scan's LDS footprint allows per CU (sizes the number of regions) */
#endif
#define MSD_SCAN_THREADS (64 * MSD_SCAN_WAVES)
#ifndef MSD_LUT_GLOBAL
#define MSD_LUT_GLOBAL 1 /* 1: the UC8 table is read through the vector cache (34 KB, L1/L2-resident; measured as fast
                            as the LDS copy inside the full kernel, and it leaves the LDS to the wavefronts' tiles);
                            0: a copy in LDS (needs MSD_TILE 1024) */
#endif
#ifndef MSD_SLICER_NG
#define MSD_SLICER_NG 3 /* bit groups (of five bits) one lane slices per item of step B */
#endif
#ifndef MSD_TILE
#define MSD_TILE 2048u /* scan positions per wavefront tile: 2048 (two runs of 16 per lane) or 1024 */
#endif
#ifndef MSD_TESTS_V2
#define MSD_TESTS_V2 1 /* 1: the preamble tests of the scan kernel written in full-rate instructions only (msd_kernels.hip
                          stage 2; scripts/micro/valu_issue.hip says which those are): 20 instead of 24 instructions per
                          position and two thirds of the issue cycles.  Round 3 measured no gain from it (the launch 1 us
                          shorter) and left it off; since the phases of a tile run at different s_setprio levels (round 4,
                          MSD_PRIO_*) the candidate rounds no longer queue behind the tests, the tests' own issue time
                          shows, and the launch is 2-3 % shorter with it (profiles/r04_priorities.txt, v2*). */
#endif
#define MSD_HALO_FRONT 328u     /* samples staged ahead of a tile: overlap 326 rounded up to 8 */
#define MSD_MAX_BATCH_SAMPLES (1ull << 28) /* hit positions are 28-bit, batch-relative */

/* hit record: bits 0..27 scan position relative to the batch (absolute a = batch_first + pos,
 * a = chunk*131072 + j); bits 28..30 which of the three preamble tests fired (1: phases 4,5
 * 2: phases 6,7  4: phase 8); bits 31..33 number of try records of this position; bits 34..63
 * index of the first of them in the batch's try list (they are consecutive, in phase order). */
typedef uint64_t msd_hit;
#define MSD_HIT_POS(h) ((h) & 0xFFFFFFFull)
#define MSD_HIT_MASK(h) ((unsigned)((h) >> 28) & 7u)
#define MSD_HIT_NLIVE(h) ((unsigned)((h) >> 31) & 7u)
#define MSD_HIT_TRY(h) ((uint64_t)((h) >> 34))

/* try record, 32 bytes */
typedef struct msd_try {
    uint8_t msg[14]; /* sliced bytes, uncorrected (demod_2400.c:191-209) */
    uint8_t tp;      /* trial phase 4..8 */
    uint8_t errbit;  /* (first) corrected bit from the syndrome table, 0xff = none */
    uint32_t addr;   /* address the score tests: CRC for AP formats, (corrected) AA otherwise */
    uint32_t crc;    /* modesChecksum of the uncorrected message */
    uint32_t pos;    /* batch-relative scan position (same as the owning hit's) */
    uint8_t errbit2; /* second corrected bit (--aggressive only), 0xff = none; errbit < errbit2 */
    uint8_t again[3]; /* msg[0], tp, errbit once more: everything a score needs is in the record's second half */
} msd_try;

/* Mode A/C candidate: every f1_sample that passes all tests of demod_2400.c:581-668; only the
 * 69-sample skip-ahead (:705) is left to the resolve stage. */
typedef struct msd_ac_hit {
    uint64_t pos;      /* batch-relative: buffer*131072 + f1_sample */
    uint32_t f2_clock; /* 60 MHz, relative to the buffer start (demod_2400.c:604) */
    uint32_t modeac;   /* demod_2400.c:672-685 */
} msd_ac_hit;

/* per persistent workgroup bookkeeping written by the scan kernel */
typedef struct msd_wg_counts {
    uint32_t nhits;
    uint32_t ntries;
    uint32_t overflow;
    uint32_t pad;
} msd_wg_counts;

/* The scan kernel is wave-autonomous: every wavefront owns a contiguous run of tiles (a "region") and
 * its own slice of the candidate arenas.  Per region: the counts, and where the region starts within its
 * workgroup's output (the workgroup's last act); per workgroup: the totals the gather kernel sums. */
typedef struct msd_region_counts {
    uint32_t nhits, ntries, overflow;
    uint32_t hbase, tbase; /* hits / tries of the workgroup's earlier regions */
    uint32_t pad[3];
} msd_region_counts;
typedef struct msd_wg_totals {
    uint32_t nhits, ntries, overflow, pad;
} msd_wg_totals;

/* ---- GPU resolve stage (msd_resolve_kernels.hip) ---- */
#define MSD_RB_MSG_CAP 1024u   /* accepted Mode S messages of one buffer: at most 131072/135 = 970 */
#define MSD_RB_AC_CAP 2048u    /* accepted Mode A/C replies of one buffer: at most 131072/70 = 1872 */
#define MSD_RB_ADD_INLINE 224u /* icaoFilterAdd addresses of one buffer reported inline: those that are not in the
                                  snapshot's active table already (the others cannot change the filter) */
#define MSD_SNAP_WORDS 16400u  /* a filter snapshot on the device: slot[8192][2] (the two tables interleaved), then the index of the active table */
/* Predicted adds of a batch: every address that has a CRC-clean DF17 / DF11(II=0) try somewhere in it, with the
 * first buffer holding such a try (the scan kernel notes them as it finds them; it does not know the filter).  The
 * resolve kernel treats the address as known in all later buffers, so a batch in which new aircraft show up still
 * converges in one pass; the host ignores the entries whose address the filter held when the batch began, checks
 * every other prediction against the adds that really happened (a predicted message can be hidden behind another
 * one) and corrects the table. */
#define MSD_PRED_SLOTS 65536u
#define MSD_PRED_LIST 32768u /* at most this many predicted addresses; more: the host resolver takes the batch */
#define MSD_PRED_NEVER 0xFFFFFFFFu
/* The device table: MSD_PRED_SLOTS 64-bit entries (generation << 56 | address << 32 | first buffer), then a 32-bit
 * cell (generation << 24 | number of entries) and MSD_PRED_LIST slot indices.  An entry of another generation is a
 * vacant slot: the table of a pipeline slot is never wiped between batches (the scan kernel fills it while it runs,
 * so nothing could wipe it in time); the host moves to the next generation with every batch and clears the memory
 * when the 8-bit count comes round.  Open addressing, linear probing from MSD_PRED_HASH. */
#define MSD_PRED_WORDS (2u * MSD_PRED_SLOTS + 2u + MSD_PRED_LIST) /* 32-bit words */
#define MSD_PRED_GENS 255u /* generations 0..254; 0xff never */
#define MSD_PRED_HASH(addr) (((addr) * 2654435761u) >> 16)
typedef struct msd_pred_entry {
    uint32_t addr;
    uint32_t first; /* buffer of the first clean squitter */
    uint32_t slot;  /* position in the device table, for corrections */
    uint32_t pad;
} msd_pred_entry;
typedef struct msd_pred_patch {
    uint32_t slot, first;
} msd_pred_patch;

/* what the resolve kernel reports per buffer; 1 KiB */
typedef struct msd_rbuf {
    uint32_t ctr[16]; /* 0 preambles, 1 bad, 2 unknown, 3/4 accepted with 0/1 fixes, 6-10 preamble phases,
                         11-15 best phases */
    uint32_t nmsgs, nadds; /* nadds: all unique addresses added (the complete list stays in device memory) */
    uint32_t version_used; /* index of the snapshot it was resolved against */
    uint32_t fallback;     /* the buffer needs the host path (cannot happen with valid candidate lists) */
    uint64_t end_now;      /* Modes.ifile_now when the buffer is done */
    uint32_t nshort;       /* entries of adds[] below, if <= MSD_RB_ADD_INLINE; else use the complete list */
    uint32_t nac;          /* accepted Mode A/C replies (they follow the Mode S messages of the buffer) */
    uint32_t cyc[8]; /* wall-clock ticks (100 MHz) lane 0 spent per phase: setup, stage, eval, walk, count, rest */
    uint32_t adds[MSD_RB_ADD_INLINE]; /* first occurrence order */
} msd_rbuf;
/* an accepted message before it is turned into a msd_message */
typedef struct msd_acc {
    uint32_t pos; /* batch-relative scan position */
    uint32_t try_index;
    int32_t score;
    uint32_t len; /* samples its signal power is summed over: msgbits * 12 / 5 */
} msd_acc;

/* ---- host tables (msd_tables.c) ---- */
/* Tables of the bit slicer / CRC of the scan kernel, one block of dwords copied to LDS:
 *  - the message is sliced five bits at a time ("groups"): 12 samples hold exactly five bits, so bit
 *    5g + k of a try with trial phase tp = 4 + q sits at t = 99 + q + 60 g + 12 k twelfths of a sample
 *    behind pa = &m[j + 2], i.e. correlator t % 5 (demod_2400.c:73-93) at sample pa + t / 5
 *    (demod_2400.c:98-177 in closed form).  Every group evaluates each of the five correlators exactly
 *    once; which bit of the group a correlator yields, and at which sample, depends on q only:
 *      MSD_SL_QOFF[q]      five 6-bit fields, field c = byte offset of correlator c's first tap from
 *                          &pa[12 g]
 *      MSD_SL_PERM[q][x]   x = the five correlator verdicts (correlator 0 in bit 4) -> the group's
 *                          five message bits (first bit in bit 4)
 *  - modesChecksum (crc.c:67-82) is linear, so the syndrome is the xor of per-group contributions:
 *      MSD_SL_GLONG[g][v]  syndrome of a 112-bit message that is zero but for group g = v
 *      MSD_SL_GSHORT[g][v] same for a 56-bit message (groups past the end contribute nothing). */
#define MSD_SL_GLONG 0u
#define MSD_SL_GLONG_ROWS 25u
#define MSD_SL_GSHORT (MSD_SL_GLONG + 32u * MSD_SL_GLONG_ROWS)
#define MSD_SL_GSHORT_ROWS 13u
#define MSD_SL_QOFF (MSD_SL_GSHORT + 32u * MSD_SL_GSHORT_ROWS)
#define MSD_SL_PERM (MSD_SL_QOFF + 8u) /* bytes [5][32] */
#define MSD_SLICER_WORDS (MSD_SL_PERM + 40u)
#define MSD_LUT_STRIDE 136u /* folded UC8 table row pitch in u16 (bank spread) */
/* The scan kernel's own copy of the folded table (round 5): a row pitch of 256 entries makes the folded byte pair
 * fold(Q) << 8 | fold(I) the index itself -- three full-rate instructions per two samples where the 136-entry pitch needs
 * two bit-field extracts, a multiply and a shifted add per sample.  With rows 512 bytes apart the same columns of every
 * row would meet in the same bytes of a cache line (a quiet band reads the first few columns of the first few rows), so
 * row r keeps column c at c ^ MSD_LUT_SCAN_SWZ(r): an XOR of both samples' indices at once, no carries.  The table
 * lies behind uc8_folded in the same device allocation (MSD_LUT_SCAN_OFFSET entries in). */
#ifndef MSD_LUT_SCAN256
#define MSD_LUT_SCAN256 1
#endif
#ifndef MSD_LUT_SCAN_SWZ_BITS
#define MSD_LUT_SCAN_SWZ_BITS 4u /* row bits that take part */
#endif
#ifndef MSD_LUT_SCAN_SWZ_SHIFT
#define MSD_LUT_SCAN_SWZ_SHIFT 3u /* ... and the column bit they start at (8 entries = 16 bytes per step) */
#endif
#define MSD_LUT_SCAN_SWZ(row) ((((uint32_t)(row)) & ((1u << MSD_LUT_SCAN_SWZ_BITS) - 1u)) << MSD_LUT_SCAN_SWZ_SHIFT)
#define MSD_LUT_SCAN_INDEX(row, col) ((uint32_t)(row) * 256u + ((uint32_t)(col) ^ MSD_LUT_SCAN_SWZ(row)))
#define MSD_LUT_SCAN_OFFSET (128u * MSD_LUT_STRIDE)
typedef struct msd_tables {
    uint16_t uc8_folded[128 * MSD_LUT_STRIDE]; /* [fold(Q)][fold(I)] of convert.c:35-61 */
    uint16_t uc8_scan[128 * 256];              /* the same, [fold(Q)][fold(I) ^ MSD_LUT_SCAN_SWZ(fold(Q))]: directly behind uc8_folded */
    uint16_t uc8_full[65536];                  /* the reference's table, for msd_tables_selftest */
    uint32_t crc_byte[256];                    /* crc.c:42-55 */
    uint32_t syn56[51], syn112[107];           /* sorted: syndrome | bit << 24 (crc.c:184-354) */
    uint32_t nsyn56, nsyn112;
    /* the same entries for the scan kernel's step C (round 5): buckets of four, bucket = (syndrome * mul) >> (32 - lg), so
     * that modesChecksumDiagnose's exact-match search (crc.c:389-412) is one 16-byte LDS read and four compares instead
     * of a seven-step binary search; an empty slot is 0 (a zero syndrome is never looked up).  The multipliers are
     * found at table build (no bucket holds more than four).  56-bit table first (MSD_SYNH_LG56 buckets), then the 112-bit one. */
#define MSD_SYNH_LG56 5u
#define MSD_SYNH_LG112 6u
#define MSD_SYNH_WORDS (4u * ((1u << MSD_SYNH_LG56) + (1u << MSD_SYNH_LG112)))
    uint32_t synhash[MSD_SYNH_WORDS];
    uint32_t synhash_mul[2];
    uint32_t slicer[MSD_SLICER_WORDS];         /* see MSD_SL_* */
} msd_tables;
void msd_tables_build(msd_tables *t, int nfix_crc);
/* the SC16Q11 table converter's lookup table (convert.c:271-295): out[1 << (2 * bits)], bits in 1..11 */
void msd_sc16q11_table_build(int bits, uint16_t *out);
/* two-bit correction (msd_tables.c); the caller frees the table */
#define MSD_FIX2_HASH(syndrome, log2_slots) (((uint32_t)(syndrome) * 0x9E3779B1u) >> (32u - (log2_slots)))
uint64_t *msd_fix2_table(const msd_tables *t, int bits, uint32_t *log2_slots);
int msd_fix2_diagnose(int bits, uint32_t syndrome, int bit[2]);
uint32_t msd_crc24(const msd_tables *t, const uint8_t *msg, int nbits);

/* ---- ICAO filter (msd_resolve.c), icao_filter.c semantics ---- */
typedef struct msd_filter {
    uint32_t slot[2][8192];
    int active;
    uint64_t next_flip;
    uint64_t set_hash;  /* xor of a mix of every known address (identity of the membership) */
    uint32_t set_count; /* number of known addresses */
    uint32_t active_used; /* occupied slots of the active table: icaoFilterAdd gives up when it is full */
} msd_filter;

/* ---- resolve stage (msd_resolve.c) ---- */
struct msd_message;
struct msd_stats;
typedef struct msd_resolver {
    msd_filter filter;
    uint64_t ifile_now;      /* Modes.ifile_now, readsb.h:289 */
    uint64_t sample_counter; /* samples consumed so far (sdr_ifile.c:172) */
    int mode_ac;
    int threads; /* host threads of the speculative batch resolve (msd_config.resolve_threads); 0 = ncpu/8 */
    int trace;   /* MSD_RESOLVE_TRACE was set when the context was created */
    struct msd_stats *stats;
    struct msd_batch_state *batch; /* scratch of msd_resolve_batch, owned by the resolver */
} msd_resolver;

/* Receives the accepted messages of one buffer, in order.  Mode S messages still lack their signal
 * level: power_req[i] = (batch-relative position << 16) | number of samples to sum, 0 for Mode A/C;
 * buffer = index of the buffer within the batch. */
typedef void (*msd_emit_fn)(const struct msd_message *mm, const uint64_t *power_req, uint32_t count,
                            uint32_t buffer, void *user);

void msd_resolver_reset(msd_resolver *r);
void msd_resolver_reset_state(msd_resolver *r); /* filter and clocks only */
void msd_resolver_reset_stats(msd_resolver *r); /* the counters only */
void msd_resolver_free(msd_resolver *r);
/* Replays the buffers [first_chunk, first_chunk + nbuffers) of a batch.  hits/tries (Mode S) and
 * ac (Mode A/C) are the batch's ordered candidate lists; valid[i] is the i-th buffer's number of
 * new samples; ts_override, if not NULL, holds
 * (sampleTimestamp, sysTimestamp) per buffer instead of the ifile clock of sdr_ifile.c:187-190. */
void msd_resolve_batch(msd_resolver *r, uint64_t first_chunk, uint32_t nbuffers,
                       const uint32_t *valid, const msd_hit *hits,
                       uint64_t nhits, const msd_try *tries, uint64_t ntries,
                       const msd_ac_hit *ac, uint64_t nac, const uint64_t *ts_override,
                       msd_emit_fn emit, void *user);
/* Second half of demodulate2400's bookkeeping, once the signal power sums are known
 * (demod_2400.c:386-408,422-427): fills signalLevel and the power statistics, in order.
 * msgs / power are arrays of msd_message / uint64_t with the given byte strides; power_req may be NULL
 * (the length then follows from msgbits, Mode A/C from msgtype). */
void msd_resolve_power_stats(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const double *means,
                             const uint32_t *buffer, const uint64_t *side, uint64_t nmsgs);
void msd_resolve_power(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const double *means,
                       void *msgs, size_t msg_stride, const uint64_t *power_req, const uint32_t *buffer,
                       const void *power, size_t power_stride, uint64_t nmsgs);

/* header fields of a whole batch on the host (msd_fields.c); msgs[i] belongs to buffer[i] */
struct msd_fields;
void msd_fields_batch(const void *msgs, size_t msg_stride, const uint32_t *buffer, uint64_t n, struct msd_fields *out);

/* ---- host half of the GPU resolve (msd_resolve.c): the cross-buffer replay ----
 * begin: clocks, snapshot 0 (= the live filter), every buffer on the to-do list.
 * replay: after a kernel pass, walks the buffers' add lists and flip times in order and decides which
 * membership version each buffer has to see.  A buffer's short list (rb[b].adds) is enough while the
 * active table of the snapshot it used is a subset of the live one; otherwise, or when it holds more
 * than inline_adds <= MSD_RB_ADD_INLINE entries, the complete list all_adds[b][MSD_RB_MSG_CAP] is
 * needed.  Returns 0 when every buffer saw the right version, 1 when `todo` (and maybe new snapshots)
 * need another pass, -2 when all_adds is NULL but needed (fetch the lists and call again), -1 when the
 * batch must go through msd_resolve_batch instead (nothing has been committed in that case).
 * pred[0..npred) is the batch's prediction list (its `first` fields are kept up to date); the
 * corrections the device table needs before the next pass come back in patches[0..*npatches).
 * commit: counters, clocks and the filter, once replay returned 0. */
void msd_gpu_resolve_begin(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, uint64_t *ts,
                           uint32_t *snap_idx, uint32_t *todo, uint32_t *ntodo);
uint32_t msd_gpu_resolve_nsnaps(const msd_resolver *r);
const uint32_t *msd_gpu_resolve_snapshot(const msd_resolver *r, uint32_t index, uint32_t *active);
int msd_gpu_resolve_replay(msd_resolver *r, uint32_t nbuffers, const msd_rbuf *rb, const uint32_t *all_adds,
                           uint32_t inline_adds, uint32_t pass, uint32_t max_snaps, msd_pred_entry *pred, uint32_t npred,
                           msd_pred_patch *patches, uint32_t *npatches, uint32_t *snap_idx, uint32_t *todo,
                           uint32_t *ntodo);
void msd_gpu_resolve_commit_state(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const msd_rbuf *rb);
void msd_gpu_resolve_commit_stats(msd_resolver *r, uint32_t nbuffers, const uint32_t *valid, const msd_rbuf *rb);

#ifdef __cplusplus
}
#endif
#endif
