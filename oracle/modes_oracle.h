/*
 * modes_oracle.h -- CPU restatement of the readsb 2.4 MSPS Mode S / Mode A/C receive path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" for the demodulator.
 *   The reference's hot-path translation units all include readsb.h, which includes
 *   <protobuf-c/protobuf-c.h>; that header is absent from this image and the round rules forbid
 *   building the reference against a stand-in header, so the reference is unbuildable here and no
 *   oracle/_ref exists.  The reference ships no demodulator tests, captures or golden vectors
 *   (SURVEY.md section 4).  What IS pinned (tests/test_oracle_known_answers.py):
 *     - the CRC-24 (generator 0xfff409, crc.c:31) against publicly known Mode S frames and the
 *       frames recorded in SURVEY.md Appendix C, and the crc.c:309-333 table self-check;
 *     - the UC8 lookup table against a direct evaluation of the convert.c:49-56 expression;
 *     - the bit-slicing plan against the closed form t = 95 + tp + 12k;
 *     - the filter-dependent acceptance sequence of SURVEY.md Appendix C (last row).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 */
#ifndef MODES_ORACLE_H
#define MODES_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* readsb.h:98-99, readsb.c:195-198 */
#define ORC_CHUNK_SAMPLES 131072u
#define ORC_OVERLAP 326u

/* convert.h:29-31 */
enum { ORC_FMT_UC8 = 0, ORC_FMT_SC16 = 1, ORC_FMT_SC16Q11 = 2 };

/* The subset of struct modesMessage (readsb.h:340-547) that the demodulator itself determines. */
typedef struct orc_message {
    uint64_t timestampMsg;    /* demod_2400.c:358 / :695 */
    uint64_t sysTimestampMsg; /* demod_2400.c:361 / :698 */
    double signalLevel;       /* demod_2400.c:398 (0 for Mode A/C) */
    uint32_t addr;            /* mode_s.c:461,561,541 / mode_ac.c:179 */
    uint32_t crc;             /* mode_s.c:440 */
    int32_t score;            /* demod_2400.c:368 */
    uint8_t msgtype;          /* DF, or 32 for Mode A/C (mode_ac.c:171) */
    uint8_t msgbits;          /* 56 / 112 / 16 */
    uint8_t correctedbits;    /* mode_s.c:488,520 */
    uint8_t bestphase;        /* demod_2400.c:384 (4..8); 0 for Mode A/C */
    uint8_t msg[14];          /* corrected message bytes (only msgbits/8 are meaningful) */
    uint8_t iid;              /* mode_s.c:476 */
    uint8_t pad;
} orc_message;

/* stats.h:61-80 demodulator counters */
typedef struct orc_stats {
    uint64_t demod_preambles;
    uint64_t demod_rejected_bad;
    uint64_t demod_rejected_unknown_icao;
    uint64_t demod_accepted[3];
    uint64_t demod_preamblePhase[5];
    uint64_t demod_bestPhase[5];
    uint64_t demod_modeac;
    uint64_t strong_signal_count;
    uint64_t samples_processed; /* readsb.c:835, includes overlap */
    uint64_t noise_power_count;
    uint64_t signal_power_count;
    double noise_power_sum;
    double signal_power_sum;
    double peak_signal_power;
    uint64_t buffers; /* number of mag_bufs demodulated (incl. the trailing empty one) */
} orc_stats;

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(int format, int preamble_threshold, int nfix_crc, int mode_ac);
/* --dcfilter (readsb.c:486): the converters with the 1 Hz DC block (convert.c:113-213,374-423); resets its state */
/* Modes.stats_15min.samples_dropped != 0 from now on / no longer: demodulate2400 then tests preambles against
 * max(PREAMBLE_THRESHOLD_PIZERO = 75, threshold) (demod_2400.c:285-290).  The 15-minute window belongs to the host program
 * (stats.c); for --ifile input it never holds anything. */
void orc_set_recently_dropped(orc_ctx *ctx, int on);
void orc_set_dc_filter(orc_ctx *ctx, int on);
/* restate a reference built with -DSC16Q11_TABLE_BITS=bits (1..11; anything else: the float path, as without the define) */
void orc_set_sc16q11_table_bits(orc_ctx *ctx, int bits);
const uint16_t *orc_sc16q11_table(const orc_ctx *ctx);
void orc_destroy(orc_ctx *ctx);

/* Replay a whole capture exactly as `readsb --device-type ifile --ifile F --iformat X --throttle`
 * would (sdr_ifile.c:164-237 + fifo.c:166-201 + readsb.c:820-855), lossless feed.
 * Messages are appended to out[0..cap); *nout receives the total produced (may exceed cap).
 * If chunk_means != NULL it receives 2 doubles (mean_level, mean_power) per buffer, up to
 * means_cap buffers.  Returns the number of buffers processed. */
uint64_t orc_replay(orc_ctx *ctx, const void *iq, uint64_t nsamples, orc_message *out, size_t cap,
                    size_t *nout, double *chunk_means, size_t means_cap);

void orc_get_stats(const orc_ctx *ctx, orc_stats *st);

/* ---- unit-level entry points (for known-answer tests) ---- */
/* convert.c:63-111 / :215-253 / :332-370; out pointers may be NULL */
void orc_convert(orc_ctx *ctx, const void *iq, uint16_t *mag, unsigned nsamples, double *mean_level,
                 double *mean_power);
/* convert.c:35-61: returns pointer to the 65536-entry UC8 table */
const uint16_t *orc_uc8_table(void);
/* crc.c:67-82 */
uint32_t orc_checksum(const uint8_t *msg, int bits);
/* crc.c:389-412: returns number of errors (0,1,2) or -1 if uncorrectable; bit positions in bit[] */
int orc_diagnose(const orc_ctx *ctx, uint32_t syndrome, int bitlen, int bit[2]);
/* mode_s.c:311-409 */
int orc_score(orc_ctx *ctx, const uint8_t *msg, int validbits);
/* icao_filter.c:76-119 */
void orc_filter_add(orc_ctx *ctx, uint32_t addr);
int orc_filter_test(const orc_ctx *ctx, uint32_t addr);
/* demod_2400.c:98-177: slices nbytes starting at trial phase try_phase for preamble position j */
void orc_slice(const uint16_t *m, uint32_t j, int try_phase, int nbytes, uint8_t *out);
/* one buffer through demodulate2400 [+ demodulate2400AC] + icaoFilterExpire; data must hold
 * valid_length samples (overlap first). */
/* Header fields of an accepted message (first stage): the field assignments
 * of decodeModesMessage after its CRC switch (mode_s.c:557-715) except the ME / MB payload decoders,
 * and decodeModeAMessage (mode_ac.c:168-202).  Same layout as msd_fields (include/modes_hip.h). */
#define ORC_INVALID_ALTITUDE (-9999) /* readsb.h:130 */
typedef struct orc_fields {
    int32_t altitude_baro;
    uint16_t AC, ID, squawk;
    uint8_t altitude_baro_valid, altitude_baro_unit;
    uint8_t squawk_valid;
    uint8_t airground; /* readsb.pb-c.h:32-35: 0 invalid, 1 ground, 2 airborne, 3 uncertain */
    uint8_t alert, alert_valid, spi, spi_valid;
    uint8_t CA, CC, CF, DR, FS, KE, ND, RI, SL, UM, VS;
    uint8_t source, addrtype, imf;
    uint32_t addr;
    /* extended squitter (decodeExtendedSquitter, mode_s.c:1373-1474); speeds,
     * headings and movement as the integers the message carries */
    uint8_t metype, mesub;
    uint8_t cpr_valid, cpr_type, cpr_odd;
    uint8_t nic_b_valid, nic_b;
    uint8_t callsign_valid;
    char callsign[8];
    uint32_t cpr_lat, cpr_lon;
    int32_t altitude_geom;
    uint8_t altitude_geom_valid, altitude_geom_unit;
    uint8_t category, category_valid;
    uint8_t nac_v_valid, nac_v;
    uint8_t velocity_valid;
    uint8_t heading_valid;
    int16_t ew_vel, ns_vel;
    uint16_t heading_raw;
    uint8_t heading_type;
    uint8_t movement;
    uint16_t ias, tas;
    uint8_t ias_valid, tas_valid, baro_rate_valid, geom_rate_valid;
    int16_t baro_rate, geom_rate;
    int16_t geom_delta;
    uint8_t geom_delta_valid;
    uint8_t emergency_valid, emergency;
    /* ME types 29 and 31 (decodeESTargetStatus mode_s.c:1058-1249, decodeESOperationalStatus :1251-1370) */
    uint8_t nav_valid, nav_altitude_source, nav_modes, nav_heading_type;
    uint8_t acc_valid, nac_p, nic_baro, nic_a, nic_c, gva, sda, sil, sil_type;
    uint8_t cc_antenna_offset;
    uint8_t commb_format; /* commb_format_t, readsb.h:166-177 */
    uint16_t nav_heading_raw, nav_qnh_raw;
    int32_t nav_mcp_altitude, nav_fms_altitude;
    uint32_t opstatus;
    /* Comm-B (decodeCommB, comm_b.c) */
    int16_t roll_q, track_rate_q;
    uint16_t gs, mach_raw;
    uint8_t commb_valid;
    uint8_t pad2[3];
} orc_fields;
/* From now on orc_replay / orc_demod_buffer also write the fields of message i to fields[i] (i < cap). */
void orc_set_fields_out(orc_ctx *ctx, orc_fields *fields, size_t cap);
/* the fields of one accepted Mode S message (msgtype 0..31) */
void orc_fields_of(const orc_message *mm, orc_fields *out);
/* The float-valued members of struct modesMessage (readsb.h:423-438,533-534: gs.v0/v2/selected, heading,
 * track_rate, roll, mach, nav.qnh, nav.heading), computed where and how the reference computes them
 * (mode_s.c:831-843,853,913-922,1131,1212,1219; comm_b.c:323-326,469-518,623-651).  Same layout as
 * msd_fields_float. */
typedef struct orc_fields_float {
    float gs_v0, gs_v2, gs_selected;
    float heading;
    float track_rate;
    float roll;
    float nav_qnh;
    float nav_heading;
    double mach;
    uint8_t gs_valid, heading_valid, heading_type, track_rate_valid, roll_valid, mach_valid, nav_qnh_valid,
        nav_heading_valid;
} orc_fields_float;
void orc_fields_float_of(const orc_message *mm, orc_fields_float *out);
/* single pieces, for known-answer tests */
int orc_decode_ac13(unsigned ac13, int *unit);
unsigned orc_decode_id13(unsigned id13);
int orc_mode_a_to_mode_c(unsigned mode_a);

/* Wire formats of accepted messages, restated from modesSendRawOutput (net_io.c:870-896) and
 * modesSendBeastOutput (net_io.c:769-835); return the number of bytes written. */
size_t orc_avr_line(const orc_message *mm, int mlat, char *out /* >= 48 */);
size_t orc_beast_frame(const orc_message *mm, uint8_t *out /* >= 44 */);

void orc_demod_buffer(orc_ctx *ctx, const uint16_t *data, unsigned valid_length,
                      uint64_t sample_timestamp, uint64_t sys_timestamp, double mean_level,
                      double mean_power, orc_message *out, size_t cap, size_t *nout);

#ifdef __cplusplus
}
#endif
#endif
