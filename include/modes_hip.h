/*
 * modes_hip.h -- C-ABI of the MI355X Mode S / Mode A/C receive path (libmodes_hip.so).
 *
 * This is the drop-in boundary for readsb's 2.4 MSPS hot path.  Every entry point names the
 * reference interface (file:line under the readsb-protobuf tree) it replaces.  Plain C types
 * only; one context per receiver / GPU / host thread; no shared mutable globals (the reference
 * keeps this state in file statics: readsb.c:60 `Modes`, convert.c:33, icao_filter.c:38-40,
 * crc.c:84-88).  All functions return 0 or a negative errno-style code, never throw, and never
 * print (MSD_CFG_TRACE and -DMSD_KERNEL_TIMING builds aside, which write timings to stderr for experiments), and never
 * read the environment: every switch is a field of the context's msd_config; the last error text of a context is available from
 * msd_last_error(ctx), the reason of the calling thread's last failed msd_create from msd_last_error(NULL).
 * After a batch could not be finished (msd_collect / msd_submit_* returned a negative code) the context
 * accepts msd_reset() and msd_destroy() only; a failed msd_launch_* consumed nothing (samples reported
 * with msd_note_dropped and a pending msd_restart still apply to the next launch).
 *
 * The library needs an AMD GPU (gfx950) at run time.  There is no CPU fallback: msd_create()
 * fails with -ENODEV when no device is present.
 */
#ifndef MODES_HIP_H
#define MODES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_CHUNK_SAMPLES 131072u /* MODES_MAG_BUF_SAMPLES, readsb.h:98-99 */
#define MSD_OVERLAP 326u          /* Modes.trailing_samples, readsb.c:198 */

/* input_format_t, convert.h:29-31 (same numbering); MSD_FMT_MAG16 = already-converted u16
 * magnitudes, i.e. the contents of struct mag_buf.data (fifo.h:57-73). */
enum { MSD_FMT_UC8 = 0, MSD_FMT_SC16 = 1, MSD_FMT_SC16Q11 = 2, MSD_FMT_MAG16 = 3 };

/* The receiver options that reach the hot path ("Config / flags"). */
typedef struct msd_config {
    int32_t device;             /* HIP device ordinal */
    int32_t format;             /* --iformat, sdr_ifile.c:88-101 */
    int32_t preamble_threshold; /* --preamble-threshold, readsb.c:503-505 (default 58) */
    int32_t nfix_crc;           /* --no-fix = 0, --fix = 1 (readsb.c:491-496), --aggressive = 2 (readsb.c:542: two-bit
                                   correction of DF17/18 against the (2, 4) tables of crc.c:374-379) */
    int32_t mode_ac;            /* --modeac, readsb.c:509-512 */
    int32_t flags;              /* MSD_CFG_* */
    uint64_t max_batch_samples; /* largest msd_submit_* call; 0 = one chunk.  Device memory of a context: about 48 bytes
                                   per sample of this (the region slices of four pipeline slots; 6.4 GB at 128 Mi samples)
                                   on the default path -- batches of four buffers or more through msd_launch_*, resolved on
                                   the GPU; the layouts without region slices (small batches, the mag_buf entry, the host
                                   resolver, MSD_CFG_NO_LEAN) add 60 bytes per sample when first used; a quarter of either
                                   with test_arena_permille = 1000; slices that overflow grow (msd_timing.reruns) */
    void *stream;               /* hipStream_t to launch on; NULL = the context creates one */
    /* ---- tuning and test settings of THIS context (0 = default); nothing in the library reads the environment ---- */
    int32_t resolve_threads;      /* host threads of the buffer-parallel resolve when a batch is resolved on the host;
                                     0 = an eighth of the CPUs, 4..64 */
    int32_t test_arena_permille;  /* candidate arenas at this many thousandths of their base size (one hit per 8 samples,
                                     one live try per 16, of a region of the scan); 0 = the default, 4000.
                                     Tests provoke the overflow path (a batch rescanned in pieces) with small values; a
                                     receiver short of device memory can run at 1000 */
    int32_t test_inline_adds;     /* tests: at most this many entries in a buffer's short add list (< MSD_RB_ADD_INLINE);
                                     0 = all of them, negative = none (every add through the long list) */
    int32_t debug_flags;          /* kernel ablations for timing experiments (results are then incomplete): 1 stop after the
                                     preamble tests, 2 after the conversion, 4 no step B, 64 / 128 record writers without
                                     their stores / altogether */
    int32_t sc16q11_table_bits;   /* MSD_FMT_SC16Q11 only: what a reference built with -DSC16Q11_TABLE_BITS=n does (debian/rules
                                     sets 8 on armhf) -- convert_sc16q11_table, convert.c:264-328: magnitudes from a
                                     2^(2n)-entry table of the top n of 11 bits of |I| and |Q|, integer sums.  1..11; 0 = the
                                     float path (convert_sc16q11_nodc).  Ignored with MSD_CFG_DC_FILTER, as the reference's
                                     selection does (convert.c:425-444) */
    int32_t reserved0;            /* 0 */
    double sample_rate;           /* MSD_CFG_DC_FILTER: Modes.sample_rate for the DC block's constant, dc_b = exp(-2 pi / rate),
                                     convert.c:479-482 (init_converter's argument; readsb.c:195 sets 2.4e6).  0 = 2 400 000.
                                     The demodulator itself is demodulate2400: its timestamps assume 2.4 MSPS whatever this says */
} msd_config;

/* The part of struct modesMessage (readsb.h:340-547) the demodulator determines; this is what
 * the reference hands to useModesMessage() (demod_2400.c:419,703; mode_s.c:2146). */
typedef struct msd_message {
    uint64_t timestampMsg;    /* 12 MHz, demod_2400.c:358 / :695 */
    uint64_t sysTimestampMsg; /* ms, demod_2400.c:361 / :698 (startup_time taken as 0) */
    double signalLevel;       /* demod_2400.c:398; 0 for Mode A/C */
    uint32_t addr;
    uint32_t crc;
    int32_t score;
    uint8_t msgtype; /* DF; 32 = Mode A/C (mode_ac.c:171) */
    uint8_t msgbits; /* 56 / 112 / 16 */
    uint8_t correctedbits;
    uint8_t bestphase; /* 4..8 (demod_2400.c:384); 0 for Mode A/C */
    uint8_t msg[14];
    uint8_t iid;
    uint8_t pad;
} msd_message;

/* msd_config.flags */
/* Stream layout of a context.  The defaults are the measured best per configuration; the switches exist
 * so that every layout stays tested (tests/test_gpu_configs.py) and measurable (profiles/). */
#define MSD_CFG_HOST_RESOLVE (1 << 4)      /* the ordered resolve stage on host threads instead of the GPU */
#define MSD_CFG_CHAIN_IN_ORDER (1 << 5)    /* resolve chain in order on the scan stream (default for UC8 / magnitudes, Mode S) */
#define MSD_CFG_CHAIN_SIDE_STREAMS (1 << 6) /* ... on side streams (default with 16-bit IQ, Mode A/C, --dcfilter) */
#define MSD_CFG_NO_LEAN (1 << 7)           /* gather kernel + dense candidate lists instead of region slices read in place */
#define MSD_CFG_NO_RESOLVE_AHEAD (1 << 8)  /* a batch is resolved in its own msd_collect only */
#define MSD_CFG_POWER_KERNEL (1 << 9)      /* signal power in a kernel of its own (default on side streams) */
#define MSD_CFG_POWER_IN_RESOLVE (1 << 10) /* ... at the end of the resolve workgroups (default in order) */
#define MSD_CFG_EMIT_KERNEL (1 << 11)      /* message records by a kernel of their own, not by the next scan's wavefronts */
#define MSD_CFG_WAIT_INPUTS_ON_STREAM (1 << 12) /* the resolve stream waits for the snapshot upload, not the caller */
#define MSD_CFG_NO_HELPER (1 << 13)        /* no helper thread: the per-message half of a batch on the calling thread */
#define MSD_CFG_REPASS_AUX (1 << 14)       /* repeated resolve passes on the high-priority side stream */
#define MSD_CFG_RECORDS_DMA (1 << 15)      /* message records (and, with MSD_CFG_DECODE_FIELDS, the field records) fetched with a copy instead of written by the kernels; measured slower on this stack */
#define MSD_CFG_TRACE (1 << 16)            /* per-batch host timings on stderr (experiments) */
#define MSD_CFG_NO_ARENA_GROWTH (1 << 17)  /* a batch that overflows the region slices of its slot is not given bigger ones and
                                              scanned again (grow_and_rescan): it goes through rerun_in_pieces and the host
                                              resolver at once, as before round 5 and as a device without spare memory does */
#define MSD_CFG_DC_SEQUENTIAL (1 << 18)    /* MSD_CFG_DC_FILTER: the DC block by the in-order kernel alone (130 Msamples/s), not by the
                                              exact parallel-in-time kernels in front of it (round 6; experiments and tests) */
#define MSD_CFG_DC_ONE_PASS (1 << 19)      /* ... one parallel pass queued instead of 24: batches of more than two blocks are
                                              not exact by then and take the in-order kernel behind the passes (tests of that path) */
#define MSD_CFG_DC_FUSED_LAUNCH (1 << 20)  /* ... all passes in ONE cooperative launch (the evaluation of pass p + 1 follows the walk of
                                              pass p block by block, nothing is launched per pass) where the batch's blocks can be
                                              resident together.  Exact like the default (two launches per pass, 24 queued) and
                                              measured slower at every batch size -- the cooperative launch costs more than the
                                              passes' launches: the switch stays off */
#define MSD_CFG_DECODE_FIELDS 1 /* also decode the header fields of every accepted message (msd_collect_fields) */
#define MSD_CFG_DC_FILTER 2     /* --dcfilter (readsb.c:486): the converters with the 1 Hz DC block (convert.c:113-213,
                                   374-423).  A FUNCTIONAL mode, not a fast one: the filter state and the float sums
                                   run through the stream strictly in order -- one dependent float chain per channel
                                   -- which bounds it at 0.13 Gsamples/s measured (55x real time for one receiver;
                                   one host core runs the same recurrence about five times faster).  It exists so
                                   that the option is there behind the same stream interface; not with
                                   MSD_FMT_MAG16 */

/* Header fields of an accepted message: what decodeModesMessage assigns after its CRC switch without
 * looking into the ME / MB payloads (mode_s.c:557-715, decodeAC13Field / decodeID13Field :101-183), and
 * decodeModeAMessage for Mode A/C replies (mode_ac.c:168-202).  Unset fields are 0. */
#define MSD_INVALID_ALTITUDE (-9999) /* readsb.h:130 */
#define MSD_NON_ICAO_ADDRESS (1u << 24) /* readsb.h:197 */
typedef struct msd_fields { /* 140 bytes */
    int32_t altitude_baro;       /* feet; meaningful with altitude_baro_valid */
    uint16_t AC;                 /* 13-bit altitude code (DF0/4/16/20) */
    uint16_t ID;                 /* 13-bit identity code (DF5/21) */
    uint16_t squawk;             /* four octal digits, hex-coded (DF5/21, Mode A/C, ES types 23 and 28) */
    uint8_t altitude_baro_valid;
    uint8_t altitude_baro_unit;  /* 0 feet, 1 metres (never decoded, mode_s.c:178-182) */
    uint8_t squawk_valid;
    uint8_t airground;           /* readsb.pb-c.h:32-35: 0 not set, 1 ground, 2 airborne, 3 uncertain */
    uint8_t alert, alert_valid, spi, spi_valid;
    uint8_t CA, CC, CF, DR, FS, KE, ND, RI, SL, UM, VS;
    uint8_t source;              /* datasource_t, readsb.h:133-142: 1 Mode A/C, 3 Mode S, 4 Mode S checked, 5 TIS-B,
                                    6 ADS-R, 7 ADS-B */
    uint8_t addrtype;            /* AIRCRAFT_META__ADDR_TYPE, readsb.pb-c.h:45-81 */
    uint8_t imf;                 /* setIMF was applied (mode_s.c:770-792) */
    uint32_t addr;               /* msd_message.addr, with MSD_NON_ICAO_ADDRESS where DF18 / IMF / Mode A/C say so */
    /* ---- extended squitter payload, DF17/18 (mode_s.c:736-1058,1373-1474): identification, positions,
     * velocity, test and status messages, target state and operational status.
     * Speeds, headings and movement are delivered as the integers the message carries (the reference
     * turns them into floats with sqrtf / atan2 / fixed tables). ---- */
    uint8_t metype, mesub;
    uint8_t cpr_valid, cpr_type, cpr_odd; /* cpr_type_t, readsb.h:155: 0 surface, 1 airborne */
    uint8_t nic_b_valid, nic_b;
    uint8_t callsign_valid;
    char callsign[8];            /* not NUL-terminated */
    uint32_t cpr_lat, cpr_lon;   /* 17 bits each */
    int32_t altitude_geom;
    uint8_t altitude_geom_valid, altitude_geom_unit;
    uint8_t category, category_valid;
    uint8_t nac_v_valid, nac_v;
    uint8_t velocity_valid;      /* ew_vel / ns_vel hold the signed components (knots, x4 already applied for subtype 2) */
    uint8_t heading_valid;       /* heading_raw / heading_type: type 19 subtypes 3,4 (x 360/1024) or surface (x 360/128) */
    int16_t ew_vel, ns_vel;
    uint16_t heading_raw;
    uint8_t heading_type;        /* heading_type_t, readsb.h:158-165; ground track from ew/ns is not derived here */
    uint8_t movement;            /* surface movement code 1..124, 0 = not available (mode_s.c:910-915) */
    uint16_t ias, tas;
    uint8_t ias_valid, tas_valid, baro_rate_valid, geom_rate_valid;
    int16_t baro_rate, geom_rate; /* ft/min */
    int16_t geom_delta;          /* ft */
    uint8_t geom_delta_valid;
    uint8_t emergency_valid, emergency; /* ES type 28 subtype 1, type 29 version 1 */
    /* ---- ME type 29, target state and status (mode_s.c:1058-1249), and type 31, aircraft operational
     * status (:1251-1370).  Headings, QNH and the antenna offset stay the integers the message carries. ---- */
    uint8_t nav_valid;           /* MSD_NAV_*: which of the nav_* values below were sent */
    uint8_t nav_altitude_source; /* nav_altitude_source_t, readsb.h:189-195: 0 invalid, 1 unknown, 2 aircraft, 3 MCP, 4 FMS */
    uint8_t nav_modes;           /* nav_modes_t, readsb.h:180-187: 1 autopilot, 2 VNAV, 4 altitude hold, 8 approach, 16 LNAV, 32 TCAS */
    uint8_t nav_heading_type;    /* heading_type_t */
    uint8_t acc_valid;           /* MSD_ACC_*: which of nac_p .. sda were sent (sil counts when sil_type != 0) */
    uint8_t nac_p, nic_baro, nic_a, nic_c, gva, sda, sil;
    uint8_t sil_type;            /* AIRCRAFT_META__SIL_TYPE, readsb.pb-c.h:101-106: 0 invalid, 1 unknown, 2 per sample, 3 per hour */
    uint8_t cc_antenna_offset;   /* operational status v2, surface: ME bits 33-40 */
    uint8_t commb_format;        /* DF20/21: commb_format_t, readsb.h:166-177: 0 unknown, 1 ambiguous, 2 empty response,
                                    3 datalink caps (BDS 1,0), 4 GICB caps (1,7), 5 aircraft ident (2,0), 6 ACAS RA (3,0),
                                    7 vertical intent (4,0), 8 track and turn (5,0), 9 heading and speed (6,0) */
    uint16_t nav_heading_raw;    /* degrees as sent (version 1 layout) or x 180/256 with MSD_NAV_HEADING_V2 */
    uint16_t nav_qnh_raw;        /* 800 + (raw - 1) * 0.8 hPa */
    int32_t nav_mcp_altitude, nav_fms_altitude; /* feet */
    uint32_t opstatus;           /* MSD_OPS_* */
    /* ---- Comm-B (DF20/21 MB field, decodeCommB comm_b.c:50-744): the register is inferred by scoring.
     * BDS 2,0 fills callsign; 4,0 nav_mcp_altitude / nav_fms_altitude / nav_qnh_raw (MSD_NAV_QNH_COMMB:
     * 800 + raw * 0.1 hPa) / nav_modes / nav_altitude_source; 5,0 heading_raw (x 90/512 deg, ground track),
     * tas and the four below; 6,0 heading_raw (magnetic), ias, baro_rate, geom_rate (the inertial rate) and mach ---- */
    int16_t roll_q;              /* roll = roll_q * 45 / 256 degrees */
    int16_t track_rate_q;        /* track angle rate = track_rate_q / 32 degrees per second */
    uint16_t gs;                 /* ground speed, knots */
    uint16_t mach_raw;           /* Mach = mach_raw * 2.048 / 512 */
    uint8_t commb_valid;         /* MSD_COMMB_* */
    uint8_t pad2[3];
} msd_fields;
/* The float-valued members of struct modesMessage (readsb.h:423-438,533-534) that decodeModesMessage derives from
 * the integers above, with the reference's own expressions evaluated on the host (msd_fields_to_float):
 *   gs.v0 / gs.v2 / gs.selected   sqrtf(ns^2 + ew^2 + 0.5) (mode_s.c:831), the surface movement tables
 *                                 (mode_s.c:216-259,913-915), BDS 5,0 ground speed (comm_b.c:577)
 *   heading                       atan2 ground track (mode_s.c:835-839), raw * 360/1024 (:853), * 360/128 (:922),
 *                                 BDS 5,0 track / BDS 6,0 heading (comm_b.c:485-490,623-628)
 *   roll, track_rate, mach        comm_b.c:469-474,513-518,649-651
 *   nav.qnh, nav.heading          mode_s.c:1131,1212,1219; comm_b.c:323-326
 * A *_valid of 0 leaves the value 0.  The GPU delivers the integers (a float square root or atan2 evaluated
 * there would not be the host libm's); this is the last step to a complete struct modesMessage. */
typedef struct msd_fields_float {
    float gs_v0, gs_v2, gs_selected;
    float heading;     /* with heading_type from msd_fields, or HEADING_GROUND_TRACK (1) when derived from ew/ns */
    float track_rate;
    float roll;
    float nav_qnh;
    float nav_heading;
    double mach;
    uint8_t gs_valid, heading_valid, heading_type, track_rate_valid, roll_valid, mach_valid, nav_qnh_valid,
        nav_heading_valid;
} msd_fields_float;

#define MSD_COMMB_ROLL 1u
#define MSD_COMMB_GS 2u
#define MSD_COMMB_TRACK_RATE 4u
#define MSD_COMMB_MACH 8u
#define MSD_NAV_QNH_COMMB 64u
#define MSD_NAV_MODES 1u
#define MSD_NAV_HEADING 2u
#define MSD_NAV_MCP_ALTITUDE 4u
#define MSD_NAV_FMS_ALTITUDE 8u
#define MSD_NAV_QNH 16u
#define MSD_NAV_HEADING_V2 32u
#define MSD_ACC_NAC_P 1u
#define MSD_ACC_NIC_BARO 2u
#define MSD_ACC_NIC_A 4u
#define MSD_ACC_NIC_C 8u
#define MSD_ACC_GVA 16u
#define MSD_ACC_SDA 32u
/* msd_fields.opstatus, struct modesMessage.opstatus (readsb.h:492-524): bit 0 valid, 1-3 version, then one bit each */
#define MSD_OPS_VALID 1u
#define MSD_OPS_VERSION(x) (((x) >> 1) & 7u)
#define MSD_OPS_OM_ACAS_RA (1u << 4)
#define MSD_OPS_OM_IDENT (1u << 5)
#define MSD_OPS_OM_ATC (1u << 6)
#define MSD_OPS_OM_SAF (1u << 7)
#define MSD_OPS_CC_ACAS (1u << 8)
#define MSD_OPS_CC_CDTI (1u << 9)
#define MSD_OPS_CC_1090_IN (1u << 10)
#define MSD_OPS_CC_ARV (1u << 11)
#define MSD_OPS_CC_TS (1u << 12)
#define MSD_OPS_CC_TC(x) (((x) >> 13) & 3u)
#define MSD_OPS_CC_UAT_IN (1u << 15)
#define MSD_OPS_CC_POA (1u << 16)
#define MSD_OPS_CC_B2_LOW (1u << 17)
#define MSD_OPS_CC_LW_VALID (1u << 18)
#define MSD_OPS_CC_LW(x) (((x) >> 19) & 15u)
#define MSD_OPS_HRD(x) (((x) >> 23) & 7u) /* heading_type_t */
#define MSD_OPS_TAH(x) (((x) >> 26) & 7u) /* heading_type_t */

/* struct stats demodulator counters, stats.h:61-80 */
typedef struct msd_stats {
    uint64_t demod_preambles;
    uint64_t demod_rejected_bad;
    uint64_t demod_rejected_unknown_icao;
    uint64_t demod_accepted[3];
    uint64_t demod_preamblePhase[5];
    uint64_t demod_bestPhase[5];
    uint64_t demod_modeac;
    uint64_t strong_signal_count;
    uint64_t samples_processed;
    uint64_t noise_power_count;
    uint64_t signal_power_count;
    double noise_power_sum;
    double signal_power_sum;
    double peak_signal_power;
    uint64_t buffers;
    uint64_t samples_dropped; /* stats.h:68; fed by msd_note_dropped() */
} msd_stats;

/* Timing of the most recent batch, measured with HIP events on the context's stream. */
typedef struct msd_timing {
    float scan_kernel_ms;   /* the fused convert+scan+slice+CRC kernel */
    float other_kernels_ms; /* compaction (+ Mode A/C, + float means) */
    float d2h_ms;
    float resolve_ms; /* ordered resolve stage, wall clock on the calling thread */
    uint64_t hits;    /* preamble positions reported by the GPU */
    uint64_t tries;   /* state-dependent candidate records reported by the GPU */
    uint64_t reruns;  /* batches re-run in halves because a candidate arena overflowed */
    uint64_t resolve_passes;   /* passes of the GPU resolve kernel over this batch; 0 = resolved on the host */
    uint64_t resolve_fallback; /* batches the GPU resolve handed to the host resolver (since msd_reset) */
    uint64_t resolve_long_lists; /* passes that had to fetch the complete per-buffer add lists (since msd_reset) */
    uint64_t timed_batches;    /* batches whose kernel times were measured (msd_set_timing_interval); the *_kernel_ms
                                  fields are those of the most recent one */
} msd_timing;

typedef struct msd_ctx msd_ctx;
/* useModesMessage-shaped sink (mode_s.h:37): called synchronously, in order, on the calling
 * thread; the message is only valid during the call. */
typedef void (*msd_message_fn)(const msd_message *mm, void *user);

/* A ready-made sink that appends to a caller-owned array (count keeps counting past cap). */
typedef struct msd_array_sink_state {
    msd_message *out;
    size_t cap;
    size_t count;
} msd_array_sink_state;
void msd_array_sink(const msd_message *mm, void *state /* msd_array_sink_state* */);
typedef struct msd_array_fields_sink_state {
    msd_message *out;
    msd_fields *fields;
    size_t cap;
    size_t count;
} msd_array_fields_sink_state;
void msd_array_fields_sink(const msd_message *mm, const msd_fields *fields, void *state);

/* ---- life cycle: replaces modesInit's modesChecksumInit/icaoFilterInit (readsb.c:241-243) and
 *      init_converter (convert.h:40-43) ---- */
int msd_create(const msd_config *cfg, msd_ctx **out);
void msd_destroy(msd_ctx *ctx);
const char *msd_last_error(const msd_ctx *ctx);

/* ---- streaming form of ifileRun + the consumer loop (sdr_ifile.c:164-237, readsb.c:820-855).
 * A capture is fed in order, in batches that are whole multiples of MSD_CHUNK_SAMPLES except the
 * last one (`last` != 0), which also produces the reference's end-of-file behaviour (a final
 * short or empty buffer).  The IQ bytes live in device memory
 * (msd_submit_device) or host memory (msd_submit_host copies them over PCIe first).
 * Messages are delivered to `sink` in the reference's order before the call returns. ---- */
int msd_submit_device(msd_ctx *ctx, const void *d_iq, uint64_t nsamples, int last,
                      msd_message_fn sink, void *user);
int msd_submit_host(msd_ctx *ctx, const void *h_iq, uint64_t nsamples, int last,
                    msd_message_fn sink, void *user);
/* Forget the stream position, ICAO filter, clock and counters (a new capture).  -EBUSY while batches are
 * outstanding. */
int msd_reset(msd_ctx *ctx);
/* The same for a receiver that replays one capture after the other: may be called as soon as the running
 * capture has been closed (last != 0) although its batches are still in flight; the batches launched
 * afterwards belong to the new capture, whose filter, clock and counters start over when the first of them
 * is collected -- msd_get_stats() between the last collect of the old capture and the first of the new one
 * still returns the old capture's counters.  Keeps the GPU busy across the boundary (bench.py). */
int msd_restart(msd_ctx *ctx);
/* A live receiver that could not hand `nsamples` samples over (rtlsdrCallback's FIFO-full branch,
 * sdr_rtlsdr.c:281-296; bladeRF the same way, sdr_bladerf.c:317-341) says so before it launches the
 * next batch.  That batch then starts with a MAGBUF_DISCONTINUOUS buffer: its 326-sample look-behind
 * is zeros instead of the end of the previous batch (fifo.c:178-181), the sample clock has advanced
 * by the dropped samples (sampleCounter, sdr_rtlsdr.c:284,299-300) and msd_stats.samples_dropped
 * grows by them (readsb.c:836) when the batch is collected.  -EINVAL after the last batch. */
int msd_note_dropped(msd_ctx *ctx, uint64_t nsamples);
/* The kernel times in msd_timing come from three hipEventRecord calls per batch, each of which holds the
 * stream for about 5 us (1 % of a 64 Mi-sample batch): measure one batch in `every` (default 1 = all,
 * 0 = none). */
int msd_set_timing_interval(msd_ctx *ctx, uint32_t every);
/* Modes.preambleThreshold for the batches launched from now on.  The reference raises it to
 * max(PREAMBLE_THRESHOLD_PIZERO = 75, threshold) while its 15-minute statistics hold dropped samples
 * (demod_2400.c:285-290); that statistics window belongs to the host program, which calls this when
 * it opens and closes.  -EINVAL outside 1..MSD_MAX_PREAMBLE_THRESHOLD (400). */
#define MSD_MAX_PREAMBLE_THRESHOLD 400 /* --preamble-threshold is clamped to 40..400, readsb.c:503-505 */
int msd_set_preamble_threshold(msd_ctx *ctx, int threshold);

/* ---- pipelined form: launch the GPU stage for a batch and return; msd_collect() waits for the
 * oldest outstanding batch, runs the ordered resolve and delivers its messages.  At most
 * MSD_PIPELINE_DEPTH batches may be outstanding.
 * msd_collect(n) also takes batch n + 1 through its resolve passes (it waits for them: they were queued when batch
 * n's filter changes were committed) and queues those of batch n + 2, so that the resolve chain runs one batch
 * ahead of the delivery and the GPU never waits for the caller between two scans; the demodulator counters and the
 * messages of a batch still appear with its own msd_collect.  Two things can show up to two collects early, because
 * starting batch n + 2 on the GPU happens inside msd_collect(n): msd_stats.samples_dropped of a gap in front of that
 * batch (msd_note_dropped), and -- across msd_restart -- the new capture's empty filter and zero clock.  Host cost per
 * context while batches are in flight: the calling thread
 * polls for events for up to 2 ms at a time before it sleeps, and one helper thread per
 * context (started by the first batch) does the same while it copies the message records and keeps the
 * order-sensitive power statistics -- two busy threads per receiver, plus a pool that only works when a batch has to
 * be resolved on the host.  Several receivers on one host should be pinned to disjoint cores near their GPU
 * (bench.py: pin_to_gpu_local_cpus). ---- */
#define MSD_PIPELINE_DEPTH 4
int msd_launch_device(msd_ctx *ctx, const void *d_iq, uint64_t nsamples, int last);
int msd_collect(msd_ctx *ctx, msd_message_fn sink, void *user);
/* msd_collect with the header fields next to every message; the context must have been created with
 * MSD_CFG_DECODE_FIELDS (the fields then come out of the same kernel that builds the message records). */
typedef void (*msd_fields_fn)(const msd_message *mm, const msd_fields *fields, void *user);
int msd_collect_fields(msd_ctx *ctx, msd_fields_fn sink, void *user);
/* The same decode for one message on the host.  `carry`: for a Mode A/C reply, the fields of the
 * previous Mode A/C reply of the same buffer (the reference reuses one message record per buffer, so a
 * reply without altitude inherits the last one's, demod_2400.c:523-528); NULL otherwise. */
void msd_decode_fields(const msd_message *mm, const msd_fields *carry, msd_fields *out);
/* The same decoder on the GPU for n messages that came from somewhere else (a Beast feed, a recording):
 * host arrays in and out, synchronous.  Mode A/C records (msgtype 32) are decoded without a carry. */
int msd_decode_fields_device(msd_ctx *ctx, const msd_message *msgs, size_t n, msd_fields *out);
/* The same for samples in host memory -- the streaming ingest behind the reference's reader thread
 * (sdr_ifile.c:192-216, the SDR callbacks of sdr_rtlsdr.c:261-326): the upload of batch k+1 runs on a
 * copy stream while batch k is scanned.  h_iq must stay valid and unchanged until the batch has been
 * collected; for the upload to be a DMA at PCIe rate it should be page-locked (msd_host_alloc). */
int msd_launch_host(msd_ctx *ctx, const void *h_iq, uint64_t nsamples, int last);
int msd_host_alloc(msd_ctx *ctx, size_t bytes, void **out); /* page-locked host memory */
void msd_host_free(msd_ctx *ctx, void *p);
/* Page-lock memory the host program owns already -- the mag_buf FIFO's sample arrays (fifo.c:74-77 allocates them with
 * malloc), the reader's block buffer -- so that the copies behind msd_convert / msd_demodulate_magbuf are DMA transfers
 * instead of staged ones (a 256 KB block: 100 -> 50 us, and two threads' copies no longer queue behind one staging
 * buffer).  Optional: unregistered memory works, slower.  Unregister before the memory is freed. */
/* A thread's first HIP call pays for the runtime's per-thread set-up (milliseconds): a thread that will call into a
 * context can pay it before its first buffer arrives. */
int msd_thread_attach(msd_ctx *ctx);
int msd_host_register(msd_ctx *ctx, void *p, size_t bytes);
void msd_host_unregister(msd_ctx *ctx, void *p);

/* The counters of everything collected so far (the order-sensitive power statistics of the last batch are summed on
 * a helper thread after msd_collect() has returned: this call waits for them). */
int msd_get_stats(const msd_ctx *ctx, msd_stats *st);
/* The size of the context's candidate arenas in thousandths of the base size: 4000 by default, 1000 when msd_create ran
 * out of device memory at the default size and fell back to the base size (it retries once), or what
 * msd_config.test_arena_permille asked for. */
int msd_arena_permille(const msd_ctx *ctx);
int msd_get_timing(const msd_ctx *ctx, msd_timing *t);
/* MSD_CFG_DC_FILTER: how the DC block of the most recent batch (or msd_convert call) was computed.  Waits for the
 * context's stream.  out[0] = 1: by the exact parallel-in-time kernels; 0: they had not arrived at an exact state for every
 * block within the passes queued and the in-order kernel finished the batch from the first such block on (or did all of
 * it: the context is MSD_CFG_DC_SEQUENTIAL, or the batch's IQ was not 16-byte aligned); out[1] = passes that did work,
 * out[2] = blocks that had to guess (all passes), out[3] = blocks.  -EINVAL for a context without the DC filter. */
int msd_dc_filter_status(msd_ctx *ctx, uint32_t out[4]);
/* mean_level / mean_power of the buffers of the most recent batch (mag_buf.mean_level/.mean_power,
 * fifo.h:70-71): 2 doubles per buffer, up to cap buffers; returns the number of buffers. */
int msd_get_buffer_means(const msd_ctx *ctx, double *means, size_t cap);

/* msd_fields -> the float-valued members of struct modesMessage, on the host (see msd_fields_float). */
void msd_fields_to_float(const msd_fields *fields, msd_fields_float *out);

/* ---- iq_convert_fn-shaped converter (convert.h:33-38): host buffers in, host buffers out,
 * bit-identical u16 magnitudes and means for UC8 / SC16 / SC16Q11 (and the SC16Q11 table of
 * msd_config.sc16q11_table_bits).  `format` is taken from the context.  Either out pointer may be NULL
 * (convert.c:104-110).  A MSD_CFG_DC_FILTER context converts with the 1 Hz DC block (convert.c:113-213,
 * 374-423): the filter state is the context's and runs on from call to call, as struct converter_state
 * does -- use such a context as a converter only, msd_launch_* of the same context advances the same state. ---- */
int msd_convert(msd_ctx *ctx, const void *iq_data, uint16_t *mag_data, unsigned nsamples,
                double *out_mean_level, double *out_mean_power);
/* The same in two halves, for a reader that wants to read its next block while this one is converted: _begin queues
 * upload, conversion and the download into mag_data and returns; _end waits and hands out the means.  One conversion in
 * flight per context (-EBUSY); iq_data and mag_data must stay valid (and should be page-locked: msd_host_register) until
 * _end has returned. */
int msd_convert_begin(msd_ctx *ctx, const void *iq_data, uint16_t *mag_data, unsigned nsamples);
int msd_convert_end(msd_ctx *ctx, double *out_mean_level, double *out_mean_power);

/* ---- demodulate2400 / demodulate2400AC-shaped entry (demod_2400.h:37-38) on one magnitude
 * buffer laid out like struct mag_buf (fifo.h:57-73): data[0..overlap) is the previous buffer's
 * tail, data[overlap..validLength) the new samples.  Runs the Mode S demodulator, then (if
 * mode_ac) the Mode A/C one, then icaoFilterExpire (readsb.c:331), like one turn of the
 * reference's consumer loop.  Uses the context's filter/clock/counters. ---- */
int msd_demodulate_magbuf(msd_ctx *ctx, const uint16_t *data, unsigned validLength, unsigned overlap,
                          uint64_t sampleTimestamp, uint64_t sysTimestamp, double mean_level,
                          double mean_power, msd_message_fn sink, void *user);
/* Several consecutive turns of that loop in one GPU batch -- what a consumer that finds n buffers queued can do instead of
 * n calls (readsb.c:820-855 takes them one at a time; the results are the same, the round trips are paid once).  The
 * buffers must follow one another in the stream: buffer k + 1's overlap region is the end of buffer k's data (what
 * fifo_enqueue writes, fifo.c:170-182) -- a MAGBUF_DISCONTINUOUS buffer starts a call of its own --, all but the last hold
 * 131072 new samples, and n x 131072 <= msd_config.max_batch_samples.  Messages are delivered in order. */
typedef struct msd_magbuf_view {
    const uint16_t *data;
    unsigned validLength, overlap;
    uint64_t sampleTimestamp, sysTimestamp;
    double mean_level, mean_power;
} msd_magbuf_view;
int msd_demodulate_magbufs(msd_ctx *ctx, const msd_magbuf_view *bufs, unsigned n, msd_message_fn sink, void *user);

#ifdef __cplusplus
}
#endif
#endif
