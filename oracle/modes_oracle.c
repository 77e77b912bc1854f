/*
 * modes_oracle.c -- CPU restatement of the readsb 2.4 MSPS receive path (see modes_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" for the demodulator (see the header for what is
 * and is not pinned).  Written from the behaviour of the reference, not from its text: the bit
 * slicer is table-driven, the ICAO filter / CRC tables live in a context instead of file statics,
 * and the replay loop restates sdr_ifile.c + fifo.c + the readsb.c consumer loop synchronously
 * (queue depth 1, which is the lossless feed the parity definition of SURVEY.md 8(b) asks for).
 *
 * Build with: gcc -std=c11 -O2 -ffp-contract=off (the reference is -std=c11 -O2 on x86-64:
 * FLT_EVAL_METHOD == 0, no FMA contraction, correctly rounded sqrtf; Makefile:12-13).
 */
#include "modes_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */
/* ------------------------------------------------------------------------------------------ */

#define FILTER_SLOTS 8192u          /* icao_filter.c:27 */
#define FILTER_VACANT 0xFFFFFFFFu   /* icao_filter.c:42 */
#define FILTER_TTL_MS 60000u        /* icao_filter.c:30 */
#define CRC_POLY 0xfff409u          /* crc.c:31 */

struct syndrome_entry { /* struct errorinfo, crc.h:32-37 */
    uint32_t syndrome;
    int errors;  /* 1 or 2; -1 while flagged for removal */
    int bit[2];
};

struct orc_ctx {
    int format, threshold, nfix, mode_ac;
    int recently_dropped; /* Modes.stats_15min.samples_dropped != 0 (demod_2400.c:285-290): a live receiver's state, set by the caller */
    int dc_filter; /* struct converter_state, convert.c:28-33 */
    int q11_bits;  /* SC16Q11_TABLE_BITS of the build being restated (0: undefined, the float path) */
    uint16_t *q11_table;
    float dc_z1_i, dc_z1_q, dc_a, dc_b;
    /* icao_filter.c:38-40 */
    uint32_t filt[2][FILTER_SLOTS];
    int active;
    uint64_t next_flip; /* icao_filter.c:151 (function static there) */
    /* crc.c:84-88 */
    const struct syndrome_entry *tab56, *tab112; /* shared between contexts with the same nfix */
    int ntab56, ntab112;
    /* readsb.h:288-289: startup_time is fixed to 0 here */
    uint64_t ifile_now;
    orc_stats st;
    /* fifo.c:43-44 */
    uint16_t carry[ORC_OVERLAP];
    uint16_t *buf; /* one mag_buf's data, ORC_CHUNK_SAMPLES + ORC_OVERLAP samples */
    /* optional field decode next to the message list */
    orc_fields *fields_out;
    size_t fields_cap;
    orc_fields ac_mm; /* what demodulate2400AC's one message record keeps between replies (demod_2400.c:523-528) */
};

/* ------------------------------------------------------------------------------------------ */
/* IQ -> magnitude (convert.c)                                                                */
/* ------------------------------------------------------------------------------------------ */

static uint16_t g_uc8[65536];
static int g_uc8_ready;

/* convert.c:35-61.  Table slot i*256+q holds the magnitude for "i" and "q" as that loop names
 * them; a sample's slot is the little-endian u16 of its (I byte, Q byte) pair, i.e. I + 256*Q,
 * so the loop's "i" is the sample's Q byte (the table is symmetric, but keep the exact form). */
static void build_uc8_table(void)
{
    if (g_uc8_ready)
        return;
    for (int i = 0; i < 256; ++i) {
        for (int q = 0; q < 256; ++q) {
            float fi = (i - 127.5) / 127.5; /* double expression rounded to float, convert.c:49 */
            float fq = (q - 127.5) / 127.5;
            float magsq = fi * fi + fq * fq;
            if (magsq > 1)
                magsq = 1;
            float mag = sqrtf(magsq);
            g_uc8[i * 256 + q] = (uint16_t)(mag * 65535.0f + 0.5f);
        }
    }
    g_uc8_ready = 1;
}

const uint16_t *orc_uc8_table(void)
{
    build_uc8_table();
    return g_uc8;
}

/* convert.c:63-111 */
static void convert_uc8(const uint8_t *iq, uint16_t *mag, unsigned n, double *ml, double *mp)
{
    uint64_t sum_level = 0, sum_power = 0;
    for (unsigned k = 0; k < n; ++k) {
        unsigned slot = (unsigned)iq[2 * k] | ((unsigned)iq[2 * k + 1] << 8);
        uint16_t v = g_uc8[slot];
        mag[k] = v;
        sum_level += v;
        sum_power += (uint32_t)v * (uint32_t)v;
    }
    if (ml)
        *ml = sum_level / 65536.0 / n; /* convert.c:105 -- 65536, not 65535 */
    if (mp)
        *mp = sum_power / 65535.0 / 65535.0 / n; /* convert.c:109 */
}

/* convert.c:215-253 (scale 32768) and :332-370 (scale 2048): float path, no DC filter.
 * The level/power sums are sequential float accumulations and the means are float divisions. */
static void convert_s16(const uint8_t *iq, uint16_t *mag, unsigned n, float scale, double *ml,
                        double *mp)
{
    float sum_level = 0, sum_power = 0;
    for (unsigned k = 0; k < n; ++k) {
        int16_t I = (int16_t)((unsigned)iq[4 * k] | ((unsigned)iq[4 * k + 1] << 8));
        int16_t Q = (int16_t)((unsigned)iq[4 * k + 2] | ((unsigned)iq[4 * k + 3] << 8));
        float fi = I / scale;
        float fq = Q / scale;
        float magsq = fi * fi + fq * fq;
        if (magsq > 1)
            magsq = 1;
        float m = sqrtf(magsq);
        sum_power += magsq;
        sum_level += m;
        mag[k] = (uint16_t)(m * 65535.0f + 0.5f);
    }
    if (ml)
        *ml = sum_level / n;
    if (mp)
        *mp = sum_power / n;
}

/* The "generic" converters the reference picks with --dcfilter (convert.c:113-163 UC8, :165-213 SC16,
 * :374-423 SC16Q11; selection :425-451): a one-pole DC estimate per channel that runs on through the
 * whole stream (struct converter_state survives the calls), subtracted before the magnitude. */
static void convert_dc(orc_ctx *ctx, const uint8_t *iq, uint16_t *mag, unsigned n, double *ml, double *mp)
{
    float z1_i = ctx->dc_z1_i, z1_q = ctx->dc_z1_q;
    const float dc_a = ctx->dc_a, dc_b = ctx->dc_b;
    float sum_level = 0, sum_power = 0;
    for (unsigned k = 0; k < n; ++k) {
        float fi, fq;
        if (ctx->format == ORC_FMT_UC8) {
            uint8_t I = iq[2 * k], Q = iq[2 * k + 1];
            fi = (I - 127.5f) / 127.5f;
            fq = (Q - 127.5f) / 127.5f;
        } else {
            const float scale = ctx->format == ORC_FMT_SC16 ? 32768.0f : 2048.0f;
            int16_t I = (int16_t)((unsigned)iq[4 * k] | ((unsigned)iq[4 * k + 1] << 8));
            int16_t Q = (int16_t)((unsigned)iq[4 * k + 2] | ((unsigned)iq[4 * k + 3] << 8));
            fi = I / scale;
            fq = Q / scale;
        }
        z1_i = fi * dc_a + z1_i * dc_b;
        z1_q = fq * dc_a + z1_q * dc_b;
        fi -= z1_i;
        fq -= z1_q;
        float magsq = fi * fi + fq * fq;
        if (magsq > 1)
            magsq = 1;
        float m = sqrtf(magsq);
        sum_power += magsq;
        sum_level += m;
        mag[k] = (uint16_t)(m * 65535.0f + 0.5f);
    }
    ctx->dc_z1_i = z1_i;
    ctx->dc_z1_q = z1_q;
    if (ml)
        *ml = sum_level / n;
    if (mp)
        *mp = sum_power / n;
}

/* --dcfilter (readsb.c:486): init_converter's "DC block @ 1Hz" (convert.c:479-482) at 2.4 MHz */
void orc_set_recently_dropped(orc_ctx *ctx, int on)
{
    ctx->recently_dropped = on != 0;
}

void orc_set_dc_filter(orc_ctx *ctx, int on)
{
    ctx->dc_filter = on;
    ctx->dc_z1_i = ctx->dc_z1_q = 0;
    ctx->dc_b = exp(-2.0 * M_PI * 1.0 / 2400000.0);
    ctx->dc_a = 1.0 - ctx->dc_b;
}

/* convert.c:264-328, a build with -DSC16Q11_TABLE_BITS=bits (debian/rules:19 sets 8 on armhf): init_sc16q11_lookup ... */
void orc_set_sc16q11_table_bits(orc_ctx *ctx, int bits)
{
    free(ctx->q11_table);
    ctx->q11_table = NULL;
    ctx->q11_bits = 0;
    if (bits < 1 || bits > 11)
        return;
    const int lose = 11 - bits;
    ctx->q11_table = malloc(sizeof(uint16_t) * ((size_t)1 << (bits * 2)));
    if (!ctx->q11_table)
        return;
    for (int i = 0; i < 2048; i += (1 << lose)) {
        for (int q = 0; q < 2048; q += (1 << lose)) {
            float fI = i / 2048.0, fQ = q / 2048.0;
            float magsq = fI * fI + fQ * fQ;
            if (magsq > 1)
                magsq = 1;
            float mag = sqrtf(magsq);
            unsigned index = ((i >> lose) << bits) | (q >> lose);
            ctx->q11_table[index] = (uint16_t)(mag * 65535.0f + 0.5f);
        }
    }
    ctx->q11_bits = bits;
}

const uint16_t *orc_sc16q11_table(const orc_ctx *ctx)
{
    return ctx->q11_table;
}

/* ... and convert_sc16q11_table, convert.c:297-328 */
static void convert_q11_table(const orc_ctx *ctx, const uint8_t *iq, uint16_t *mag, unsigned n, double *ml, double *mp)
{
    const int bits = ctx->q11_bits, lose = 11 - bits;
    uint64_t sum_level = 0, sum_power = 0;
    for (unsigned k = 0; k < n; ++k) {
        uint16_t I = abs((int16_t)((unsigned)iq[4 * k] | ((unsigned)iq[4 * k + 1] << 8))) & 2047;
        uint16_t Q = abs((int16_t)((unsigned)iq[4 * k + 2] | ((unsigned)iq[4 * k + 3] << 8))) & 2047;
        uint16_t v = ctx->q11_table[((I >> lose) << bits) | (Q >> lose)];
        mag[k] = v;
        sum_level += v;
        sum_power += (uint32_t)v * (uint32_t)v;
    }
    if (ml)
        *ml = sum_level / 65536.0 / n;
    if (mp)
        *mp = sum_power / 65535.0 / 65535.0 / n;
}

void orc_convert(orc_ctx *ctx, const void *iq, uint16_t *mag, unsigned n, double *ml, double *mp)
{
    if (ctx->dc_filter) {
        convert_dc(ctx, iq, mag, n, ml, mp);
        return;
    }
    switch (ctx->format) { /* convert.c:425-444 selection, filter_dc == 0 */
    case ORC_FMT_UC8:
        convert_uc8(iq, mag, n, ml, mp);
        break;
    case ORC_FMT_SC16:
        convert_s16(iq, mag, n, 32768.0f, ml, mp);
        break;
    default:
        if (ctx->q11_bits) /* convert.c:437-438: the table path takes the float path's place in converters_table */
            convert_q11_table(ctx, iq, mag, n, ml, mp);
        else
            convert_s16(iq, mag, n, 2048.0f, ml, mp);
        break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* CRC-24 and single-bit syndrome tables (crc.c)                                              */
/* ------------------------------------------------------------------------------------------ */

static uint32_t g_crc_byte[256];
static uint32_t g_bit_syndrome[112];
static int g_crc_ready;

/* crc.c:67-82 */
uint32_t orc_checksum(const uint8_t *msg, int bits)
{
    int n = bits / 8;
    uint32_t rem = 0;
    for (int i = 0; i < n - 3; ++i)
        rem = ((rem << 8) ^ g_crc_byte[msg[i] ^ ((rem >> 16) & 0xff)]) & 0xffffffu;
    return rem ^ ((uint32_t)msg[n - 3] << 16) ^ ((uint32_t)msg[n - 2] << 8) ^ msg[n - 1];
}

/* crc.c:42-65 */
static void build_crc_tables(void)
{
    if (g_crc_ready)
        return;
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t c = b << 16;
        for (int k = 0; k < 8; ++k)
            c = (c & 0x800000u) ? ((c << 1) ^ CRC_POLY) : (c << 1);
        g_crc_byte[b] = c & 0xffffffu;
    }
    g_crc_ready = 1; /* orc_checksum is usable from here */
    uint8_t probe[14];
    memset(probe, 0, sizeof probe);
    for (int i = 0; i < 112; ++i) {
        probe[i >> 3] ^= (uint8_t)(0x80u >> (i & 7));
        g_bit_syndrome[i] = orc_checksum(probe, 112);
        probe[i >> 3] ^= (uint8_t)(0x80u >> (i & 7));
    }
}

static int by_syndrome(const void *a, const void *b)
{
    const struct syndrome_entry *x = a, *y = b;
    return (int)x->syndrome - (int)y->syndrome; /* crc.c:94-98 */
}

/* prepareSubtable, crc.c:133-153: every error pattern of up to max_errors bits out of [startbit, endbit),
 * extending the pattern of *base, in the order the reference's recursion visits them */
static int fill_patterns(struct syndrome_entry *t, int n, int offset, int startbit, int endbit,
                         const struct syndrome_entry *base, int have, int max_errors)
{
    if (have >= max_errors)
        return n;
    for (int i = startbit; i < endbit; ++i) {
        t[n] = *base;
        t[n].syndrome ^= g_bit_syndrome[i + offset];
        t[n].errors = have + 1;
        t[n].bit[have] = i;
        const int mine = n++;
        n = fill_patterns(t, n, offset, i + 1, endbit, &t[mine], have + 1, max_errors);
    }
    return n;
}

/* flagCollisions, crc.c:155-178: every pattern of first..last error bits whose syndrome is in the
 * table marks that entry (errors = -1) */
static int flag_collisions(struct syndrome_entry *t, int n, int offset, int startbit, int endbit, uint32_t base,
                           int nbits, int first, int last)
{
    if (nbits > last)
        return 0;
    int count = 0;
    for (int i = startbit; i < endbit; ++i) {
        struct syndrome_entry key;
        key.syndrome = base ^ g_bit_syndrome[i + offset];
        if (nbits >= first) {
            struct syndrome_entry *hit = bsearch(&key, t, (size_t)n, sizeof t[0], by_syndrome);
            if (hit && hit->errors != -1) {
                hit->errors = -1;
                ++count;
            }
        }
        count += flag_collisions(t, n, offset, i + 1, endbit, key.syndrome, nbits + 1, first, last);
    }
    return count;
}

/* prepareErrorTable, crc.c:184-354: patterns of up to max_correct wrong bits in bits 5..bits-1 (the DF
 * field is never "corrected"), syndromes taken at offset 112-bits, sorted; syndromes that more than one
 * pattern produces are dropped altogether (:236-258), and so are those that a pattern of
 * max_correct+1 .. max_detect bits produces as well (:266-287).  modesChecksumInit (crc.c:360-381) asks
 * for (1, 1) with --fix and (2, 4) with --aggressive. */
static struct syndrome_entry *build_syndrome_table(int bits, int max_correct, int max_detect, int *size_out)
{
    int maxsize = 0;
    for (int k = 1, c = 1; k <= max_correct; ++k) {
        c = c * (bits - 5 - (k - 1)) / k; /* (bits-5 choose k), crc.c:103-118 */
        maxsize += c;
    }
    struct syndrome_entry *t = malloc((size_t)maxsize * sizeof t[0]);
    struct syndrome_entry base = {0, 0, {-1, -1}};
    int n = fill_patterns(t, 0, 112 - bits, 5, bits, &base, 0, max_correct);
    qsort(t, (size_t)n, sizeof t[0], by_syndrome);
    int j = 0;
    for (int i = 0; i < n; ++i) {
        if (i + 1 < n && t[i + 1].syndrome == t[i].syndrome) {
            while (i + 1 < n && t[i + 1].syndrome == t[i].syndrome)
                ++i;
            continue; /* t[i] was the last of the duplicates */
        }
        t[j++] = t[i];
    }
    n = j;
    if (max_detect > max_correct &&
        flag_collisions(t, n, 112 - bits, 5, bits, 0, 1, max_correct + 1, max_detect) > 0) {
        j = 0;
        for (int i = 0; i < n; ++i)
            if (t[i].errors != -1)
                t[j++] = t[i];
        n = j;
    }
    *size_out = n;
    return t;
}

/* the tables only depend on nfix: built once per process and shared (the --aggressive ones take a moment) */
static const struct syndrome_entry *g_tab[3][2];
static int g_ntab[3][2];

static void syndrome_tables(int nfix, const struct syndrome_entry **t56, int *n56, const struct syndrome_entry **t112,
                            int *n112)
{
    if (!g_tab[nfix][0]) {
        const int detect = nfix == 1 ? 1 : 4; /* crc.c:367-379 */
        g_tab[nfix][0] = build_syndrome_table(56, nfix, detect, &g_ntab[nfix][0]);
        g_tab[nfix][1] = build_syndrome_table(112, nfix, detect, &g_ntab[nfix][1]);
    }
    *t56 = g_tab[nfix][0];
    *n56 = g_ntab[nfix][0];
    *t112 = g_tab[nfix][1];
    *n112 = g_ntab[nfix][1];
}

/* crc.c:389-412 */
int orc_diagnose(const orc_ctx *ctx, uint32_t syndrome, int bitlen, int bit[2])
{
    bit[0] = bit[1] = -1;
    if (syndrome == 0)
        return 0;
    const struct syndrome_entry *t = (bitlen == 56) ? ctx->tab56 : ctx->tab112;
    int n = (bitlen == 56) ? ctx->ntab56 : ctx->ntab112;
    if (n == 0)
        return -1; /* nfix_crc == 0: no table (crc.c:362-365,408-409) */
    struct syndrome_entry key = {syndrome, 0, {-1, -1}};
    const struct syndrome_entry *hit = bsearch(&key, t, (size_t)n, sizeof t[0], by_syndrome);
    if (!hit)
        return -1;
    bit[0] = hit->bit[0];
    bit[1] = hit->bit[1];
    return hit->errors;
}

/* ------------------------------------------------------------------------------------------ */
/* ICAO address filter (icao_filter.c)                                                        */
/* ------------------------------------------------------------------------------------------ */

/* icao_filter.c:44-65, Jenkins one-at-a-time over the three address bytes */
static uint32_t addr_hash(uint32_t a)
{
    uint32_t h = 0;
    for (int k = 0; k < 3; ++k) {
        h += (a >> (8 * k)) & 0xff;
        h += h << 10;
        h ^= h >> 6;
    }
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return h & (FILTER_SLOTS - 1);
}

/* icao_filter.c:76-97 */
void orc_filter_add(orc_ctx *ctx, uint32_t addr)
{
    uint32_t *t = ctx->filt[ctx->active];
    uint32_t h0 = addr_hash(addr), h = h0;
    int full = 0;
    while (t[h] != FILTER_VACANT && t[h] != addr) {
        h = (h + 1) & (FILTER_SLOTS - 1);
        if (h == h0) {
            full = 1;
            break;
        }
    }
    if (full)
        return; /* icao_filter.c:82-85 returns before the second insert */
    if (t[h] == FILTER_VACANT)
        t[h] = addr;

    /* second copy keyed by the low 16 bits (icao_filter.c:89-97) */
    uint32_t low = addr & 0xffffu;
    h0 = h = addr_hash(low);
    while (t[h] != FILTER_VACANT && (t[h] & 0xffffu) != low) {
        h = (h + 1) & (FILTER_SLOTS - 1);
        if (h == h0)
            return;
    }
    if (t[h] == FILTER_VACANT)
        t[h] = addr;
}

/* icao_filter.c:99-119: table a first, then table b, exact match */
int orc_filter_test(const orc_ctx *ctx, uint32_t addr)
{
    for (int which = 0; which < 2; ++which) {
        const uint32_t *t = ctx->filt[which];
        uint32_t h0 = addr_hash(addr), h = h0;
        while (t[h] != FILTER_VACANT && t[h] != addr) {
            h = (h + 1) & (FILTER_SLOTS - 1);
            if (h == h0)
                break;
        }
        if (t[h] == addr)
            return 1;
    }
    return 0;
}

/* icao_filter.c:150-164 with mstime() == Modes.ifile_now (util.c:61-64) */
static void filter_expire(orc_ctx *ctx)
{
    uint64_t now = ctx->ifile_now;
    if (now >= ctx->next_flip) {
        int other = ctx->active ^ 1;
        memset(ctx->filt[other], 0xFF, sizeof ctx->filt[other]);
        ctx->active = other;
        ctx->next_flip = now + FILTER_TTL_MS;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* message scoring and the acceptance part of decode (mode_s.c)                               */
/* ------------------------------------------------------------------------------------------ */

/* mode_s.h:57-102 for the one call shape the path uses: bits 9..32 (the AA field) */
static uint32_t field_aa(const uint8_t *msg)
{
    return ((uint32_t)msg[1] << 16) | ((uint32_t)msg[2] << 8) | msg[3];
}

/* mode_s.c:81-83 */
static int bits_for_df(int df)
{
    return (df & 0x10) ? 112 : 56;
}

/* mode_s.c:266-281 for at most one error bit */
static uint32_t aa_after_fix(uint32_t addr, int nerr, const int bit[2])
{
    for (int i = 0; i < nerr; ++i)
        if (bit[i] >= 8 && bit[i] <= 31)
            addr ^= 1u << (31 - bit[i]);
    return addr;
}

static int all_zero(const uint8_t *p, int n)
{
    for (int i = 0; i < n; ++i)
        if (p[i])
            return 0;
    return 1;
}

/* mode_s.c:311-409 */
int orc_score(orc_ctx *ctx, const uint8_t *msg, int validbits)
{
    if (validbits < 56)
        return -2;
    int df = msg[0] >> 3;
    int nbits = bits_for_df(df);
    if (validbits < nbits)
        return -2;
    if (all_zero(msg, nbits / 8))
        return -2;

    uint32_t crc = orc_checksum(msg, nbits);
    int bit[2], nerr;

    switch (df) {
    case 0: case 4: case 5: case 16:
    case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
        return orc_filter_test(ctx, crc) ? 1000 : -1;

    case 11: {
        int iid = (int)(crc & 0x7f);
        nerr = orc_diagnose(ctx, crc & 0xffff80u, nbits, bit);
        if (nerr < 0 || nerr > 1)
            return -2;
        uint32_t addr = aa_after_fix(field_aa(msg), nerr, bit);
        int known = orc_filter_test(ctx, addr);
        if (iid == 0)
            return (known ? 1600 : 750) / (nerr + 1);
        return known ? 1000 / (nerr + 1) : -1;
    }

    case 17: case 18: {
        nerr = orc_diagnose(ctx, crc, nbits, bit);
        if (nerr < 0)
            return -2;
        uint32_t addr = aa_after_fix(field_aa(msg), nerr, bit);
        return (orc_filter_test(ctx, addr) ? 1800 : 1400) / (nerr + 1);
    }

    case 20: case 21:
        return orc_filter_test(ctx, crc) ? 1000 : -2;

    default:
        return -2;
    }
}

/* mode_s.c:424-555 (CRC / address acceptance) and :717-726 (the only icaoFilterAdd call site).
 * Field decoding in between never rejects.  Returns 0, -1 or -2 like decodeModesMessage. */
static int decode_accept(orc_ctx *ctx, orc_message *mm, const uint8_t *raw)
{
    memcpy(mm->msg, raw, 14);
    uint8_t *msg = mm->msg;
    if (all_zero(msg, 7))
        return -2;

    int df = msg[0] >> 3;
    mm->msgtype = (uint8_t)df;
    mm->msgbits = (uint8_t)bits_for_df(df);
    mm->crc = orc_checksum(msg, mm->msgbits);
    mm->correctedbits = 0;
    mm->addr = 0;
    mm->iid = 0;
    int bit[2], nerr;

    switch (df) {
    case 0: case 4: case 5: case 16:
    case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
        if (!orc_filter_test(ctx, mm->crc))
            return -1;
        mm->addr = mm->crc;
        break;

    case 11:
        mm->iid = (uint8_t)(mm->crc & 0x7f);
        if (mm->crc & 0xffff80u) {
            nerr = orc_diagnose(ctx, mm->crc & 0xffff80u, mm->msgbits, bit);
            if (nerr < 0 || nerr > 1)
                return -2;
            mm->correctedbits = (uint8_t)nerr;
            for (int i = 0; i < nerr; ++i) /* crc.c:417-425 */
                msg[bit[i] >> 3] ^= (uint8_t)(0x80u >> (bit[i] & 7));
            if (!orc_filter_test(ctx, field_aa(msg)))
                return -1; /* corrected DF11 must match a known aircraft, mode_s.c:492-498 */
        }
        break;

    case 17: case 18:
        if (mm->crc != 0) {
            nerr = orc_diagnose(ctx, mm->crc, mm->msgbits, bit);
            if (nerr < 0)
                return -2;
            uint32_t before = field_aa(msg);
            mm->correctedbits = (uint8_t)nerr;
            for (int i = 0; i < nerr; ++i)
                msg[bit[i] >> 3] ^= (uint8_t)(0x80u >> (bit[i] & 7));
            uint32_t after = field_aa(msg);
            if (before != after && !orc_filter_test(ctx, after))
                return -1; /* mode_s.c:522-526 */
        }
        break;

    case 20: case 21:
        if (!orc_filter_test(ctx, mm->crc))
            return -1;
        mm->addr = mm->crc;
        break;

    default:
        return -2;
    }

    if (df == 11 || df == 17 || df == 18)
        mm->addr = field_aa(msg); /* mode_s.c:559-562 */

    if (!mm->correctedbits && (df == 17 || (df == 11 && mm->iid == 0)))
        orc_filter_add(ctx, mm->addr); /* mode_s.c:717-726 */
    /* the reference leaves stale bytes of an earlier trial behind a short message
     * (demod_2400.c:207-209 only writes bytelen bytes); report zeros there instead */
    memset(msg + mm->msgbits / 8, 0, (size_t)(14 - mm->msgbits / 8));
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Mode S demodulator (demod_2400.c:73-428)                                                   */
/* ------------------------------------------------------------------------------------------ */

/* demod_2400.c:73-93 */
static int correlate(const uint16_t *m, int which)
{
    switch (which) {
    case 0: return 18 * m[0] - 15 * m[1] - 3 * m[2];
    case 1: return 14 * m[0] - 5 * m[1] - 9 * m[2];
    case 2: return 16 * m[0] + 5 * m[1] - 20 * m[2];
    case 3: return 7 * m[0] + 11 * m[1] - 18 * m[2];
    default: return 4 * m[0] + 15 * m[1] - 20 * m[2] + 1 * m[3];
    }
}

/* demod_2400.c:98-177 as data: for each starting phase, the (sample offset, correlator) of the
 * eight bits of one byte, and how far the read pointer advances afterwards. */
static const struct { uint8_t off[8], cor[8], advance; } byte_plan[5] = {
    {{0, 2, 4, 7, 9, 12, 14, 16}, {0, 2, 4, 1, 3, 0, 2, 4}, 19},
    {{0, 2, 5, 7, 9, 12, 14, 17}, {1, 3, 0, 2, 4, 1, 3, 0}, 19},
    {{0, 2, 5, 7, 10, 12, 14, 17}, {2, 4, 1, 3, 0, 2, 4, 1}, 19},
    {{0, 3, 5, 7, 10, 12, 15, 17}, {3, 0, 2, 4, 1, 3, 0, 2}, 19},
    {{0, 3, 5, 8, 10, 12, 15, 17}, {4, 1, 3, 0, 2, 4, 1, 3}, 20},
};

void orc_slice(const uint16_t *m, uint32_t j, int try_phase, int nbytes, uint8_t *out)
{
    const uint16_t *p = &m[j + 19] + try_phase / 5; /* demod_2400.c:188 */
    int phase = try_phase % 5;                      /* demod_2400.c:189 */
    for (int b = 0; b < nbytes; ++b) {
        unsigned v = 0;
        for (int k = 0; k < 8; ++k)
            v = (v << 1) | (correlate(p + byte_plan[phase].off[k], byte_plan[phase].cor[k]) > 0);
        out[b] = (uint8_t)v;
        p += byte_plan[phase].advance;
        phase = (phase + 1) % 5;
    }
}

struct trial_state {
    uint8_t store[2][14];
    int cur; /* which store the next trial writes into (demod_2400.c:227 ping-pong) */
    const uint8_t *best;
    int bestscore, bestphase;
};

/* demod_2400.c:183-229 */
static void try_phase(orc_ctx *ctx, const uint16_t *m, uint32_t j, int tp, struct trial_state *ts)
{
    ctx->st.demod_preamblePhase[tp - 4]++;
    uint8_t *msg = ts->store[ts->cur];
    orc_slice(m, j, tp, 1, msg);
    int nbytes;
    switch (msg[0] >> 3) {
    case 0: case 4: case 5: case 11:
        nbytes = 7;
        break;
    case 16: case 17: case 18: case 20: case 21: case 24:
        nbytes = 14;
        break;
    default:
        nbytes = 1;
        break;
    }
    int score = -2;
    if (nbytes > 1) {
        orc_slice(m, j, tp, nbytes, msg);
        score = orc_score(ctx, msg, nbytes * 8);
    }
    if (score > ts->bestscore) { /* strict: the first-tried phase wins ties */
        ts->best = msg;
        ts->bestscore = score;
        ts->bestphase = tp;
        ts->cur ^= 1;
    }
}

/* ---------------------------------------------------------------------------------------- */
/* header fields (mode_s.c:101-183,557-715; mode_ac.c:59-202)                                */
/* ---------------------------------------------------------------------------------------- */

static unsigned getbits(const uint8_t *data, unsigned firstbit, unsigned lastbit) /* mode_s.h:57-102 */
{
    unsigned v = 0;
    for (unsigned b = firstbit; b <= lastbit; ++b) /* bit 1 = MSB of byte 0 */
        v = (v << 1) | ((data[(b - 1) >> 3] >> (7 - ((b - 1) & 7))) & 1u);
    return v;
}

unsigned orc_decode_id13(unsigned id13) /* mode_s.c:101-143 */
{
    static const struct { unsigned from, to; } map[12] = {
        {0x1000, 0x0010}, {0x0800, 0x1000}, {0x0400, 0x0020}, {0x0200, 0x2000}, {0x0100, 0x0040}, {0x0080, 0x4000},
        {0x0020, 0x0100}, {0x0010, 0x0001}, {0x0008, 0x0200}, {0x0004, 0x0002}, {0x0002, 0x0400}, {0x0001, 0x0004},
    };
    unsigned hex = 0;
    for (int i = 0; i < 12; ++i)
        if (id13 & map[i].from)
            hex |= map[i].to;
    return hex;
}

/* mode_ac.c:101-163 */
static int gillham_to_mode_c(unsigned ModeA)
{
    unsigned FiveHundreds = 0, OneHundreds = 0;
    if ((ModeA & 0xFFFF8889u) != 0 || (ModeA & 0x000000F0u) == 0)
        return ORC_INVALID_ALTITUDE;
    if (ModeA & 0x0010) OneHundreds ^= 0x007;
    if (ModeA & 0x0020) OneHundreds ^= 0x003;
    if (ModeA & 0x0040) OneHundreds ^= 0x001;
    if ((OneHundreds & 5) == 5)
        OneHundreds ^= 2;
    if (OneHundreds > 5)
        return ORC_INVALID_ALTITUDE;
    if (ModeA & 0x0002) FiveHundreds ^= 0x0FF;
    if (ModeA & 0x0004) FiveHundreds ^= 0x07F;
    if (ModeA & 0x1000) FiveHundreds ^= 0x03F;
    if (ModeA & 0x2000) FiveHundreds ^= 0x01F;
    if (ModeA & 0x4000) FiveHundreds ^= 0x00F;
    if (ModeA & 0x0100) FiveHundreds ^= 0x007;
    if (ModeA & 0x0200) FiveHundreds ^= 0x003;
    if (ModeA & 0x0400) FiveHundreds ^= 0x001;
    if (FiveHundreds & 1)
        OneHundreds = 6 - OneHundreds;
    return (int)(FiveHundreds * 5 + OneHundreds) - 13;
}

/* modeACInit + modeAToModeC (mode_ac.c:63-87) with the index packing of track.h:246-256: a table over
 * the 4096 twelve-bit codes, looked up through the packed index (so stray bits are ignored) */
static int g_mode_a_to_c[4096];
static int g_mode_a_ready;

int orc_mode_a_to_mode_c(unsigned modeA)
{
    if (!g_mode_a_ready) {
        for (unsigned i = 0; i < 4096; ++i) {
            const unsigned a = (i & 0007) | ((i & 0070) << 1) | ((i & 0700) << 2) | ((i & 07000) << 3);
            g_mode_a_to_c[i] = gillham_to_mode_c(a);
        }
        g_mode_a_ready = 1;
    }
    const unsigned i = (modeA & 0x0007) | ((modeA & 0x0070) >> 1) | ((modeA & 0x0700) >> 2) | ((modeA & 0x7000) >> 3);
    return g_mode_a_to_c[i];
}

int orc_decode_ac13(unsigned AC13Field, int *unit) /* mode_s.c:152-183 */
{
    const unsigned m_bit = AC13Field & 0x0040, q_bit = AC13Field & 0x0010;
    if (m_bit) {
        *unit = 1;
        return ORC_INVALID_ALTITUDE;
    }
    *unit = 0;
    if (q_bit) {
        const int n = (int)(((AC13Field & 0x1F80) >> 2) | ((AC13Field & 0x0020) >> 1) | (AC13Field & 0x000F));
        return n * 25 - 1000;
    }
    const int n = orc_mode_a_to_mode_c(orc_decode_id13(AC13Field));
    return n < -12 ? ORC_INVALID_ALTITUDE : 100 * n;
}

static int decode_ac12(unsigned AC12Field) /* mode_s.c:187-208 */
{
    if (AC12Field & 0x10) {
        const int n = (int)(((AC12Field & 0x0FE0) >> 1) | (AC12Field & 0x000F));
        return n * 25 - 1000;
    }
    int n = (int)(((AC12Field & 0x0FC0) << 1) | (AC12Field & 0x003F));
    n = orc_mode_a_to_mode_c(orc_decode_id13((unsigned)n));
    return n < -12 ? ORC_INVALID_ALTITUDE : 100 * n;
}

enum { AT_ADSB_ICAO, AT_ADSB_ICAO_NT, AT_ADSR_ICAO, AT_TISB_ICAO, AT_ADSB_OTHER, AT_ADSR_OTHER, AT_TISB_TRACKFILE,
       AT_TISB_OTHER, AT_MODE_A, AT_UNKNOWN };                   /* readsb.pb-c.h:45-81 */
enum { SRC_MODE_AC = 1, SRC_MODE_S = 3, SRC_MODE_S_CHECKED = 4, SRC_TISB = 5, SRC_ADSR = 6, SRC_ADSB = 7 }; /* readsb.h:133-142 */
#define NON_ICAO 0x01000000u /* readsb.h:197 */

static void set_imf(orc_fields *f) /* mode_s.c:770-792 */
{
    f->addr |= NON_ICAO;
    f->imf = 1;
    switch (f->addrtype) {
    case AT_ADSB_ICAO: case AT_ADSB_ICAO_NT: f->addrtype = AT_ADSB_OTHER; break;
    case AT_TISB_ICAO: f->addrtype = AT_TISB_TRACKFILE; break;
    case AT_ADSR_ICAO: f->addrtype = AT_ADSR_OTHER; break;
    default: break;
    }
}

static const char ais_charset[64] = "@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_ !\"#$%&'()*+,-./0123456789:;<=>?"; /* ais_charset.c */

static void es_ident(const uint8_t *me, orc_fields *f) /* mode_s.c:736-766 */
{
    f->mesub = (uint8_t)getbits(me, 6, 8);
    for (unsigned i = 0; i < 8; ++i)
        f->callsign[i] = ais_charset[getbits(me, 9 + 6 * i, 14 + 6 * i)];
    f->callsign_valid = 1;
    for (unsigned i = 0; i < 8; ++i) {
        const char c = f->callsign[i];
        if (!(c >= 'A' && c <= 'Z') && !(c >= '0' && c <= '9') && c != ' ') {
            f->callsign_valid = 0;
            break;
        }
    }
    f->category = (uint8_t)(((0x0E - f->metype) << 4) | f->mesub);
    f->category_valid = 1;
}

/* The float-valued members decodeModesMessage assigns (readsb.h:423-438,533-534), written where the reference
 * writes them, with its expressions and its variable types; orc_fields_float_of points this at its output. */
enum { HT_INVALID, HT_GROUND_TRACK, HT_TRUE, HT_MAGNETIC, HT_MAGNETIC_OR_TRUE, HT_TRACK_OR_HEADING }; /* readsb.h:158-165 */
static __thread orc_fields_float *cur_ff;

static float movement_field_v2(unsigned movement) /* mode_s.c:216-234 */
{
    if (movement >= 125) return 0;
    else if (movement == 124) return 180;
    else if (movement >= 109) return 100 + (movement - 109 + 0.5) * 5;
    else if (movement >= 94) return 70 + (movement - 94 + 0.5) * 2;
    else if (movement >= 39) return 15 + (movement - 39 + 0.5) * 1;
    else if (movement >= 13) return 2 + (movement - 13 + 0.5) * 0.50;
    else if (movement >= 9) return 1 + (movement - 9 + 0.5) * 0.25;
    else if (movement >= 3) return 0.125 + (movement - 3 + 0.5) * 0.875 / 6;
    else if (movement >= 2) return 0.125 / 2;
    else return 0;
}

static float movement_field_v0(unsigned movement) /* mode_s.c:242-259 */
{
    if (movement >= 125) return 0;
    else if (movement == 124) return 180;
    else if (movement >= 109) return 100 + (movement - 109 + 0.5) * 5;
    else if (movement >= 94) return 70 + (movement - 94 + 0.5) * 2;
    else if (movement >= 39) return 15 + (movement - 39 + 0.5) * 1;
    else if (movement >= 13) return 2 + (movement - 13 + 0.5) * 0.50;
    else if (movement >= 9) return 1 + (movement - 9 + 0.5) * 0.25;
    else if (movement >= 2) return 0.125 + (movement - 2 + 0.5) * 0.125;
    else return 0;
}

static void es_velocity(const uint8_t *me, orc_fields *f, int check_imf) /* mode_s.c:794-900 */
{
    f->mesub = (uint8_t)getbits(me, 6, 8);
    if (f->mesub < 1 || f->mesub > 4)
        return;
    if (check_imf && getbits(me, 9, 9))
        set_imf(f);
    f->nac_v_valid = 1;
    f->nac_v = (uint8_t)getbits(me, 11, 13);
    switch (f->mesub) {
    case 1: case 2: {
        const unsigned ew_raw = getbits(me, 15, 24), ns_raw = getbits(me, 26, 35);
        if (ew_raw && ns_raw) {
            const int ew_vel = (int)(ew_raw - 1) * (getbits(me, 14, 14) ? -1 : 1) * ((f->mesub == 2) ? 4 : 1);
            const int ns_vel = (int)(ns_raw - 1) * (getbits(me, 25, 25) ? -1 : 1) * ((f->mesub == 2) ? 4 : 1);
            f->ew_vel = (int16_t)ew_vel; /* gs = sqrtf(ns^2 + ew^2 + 0.5), track = atan2(ew, ns): left to the caller */
            f->ns_vel = (int16_t)ns_vel;
            f->velocity_valid = 1;
            if (cur_ff) { /* mode_s.c:830-843 */
                cur_ff->gs_v0 = cur_ff->gs_v2 = cur_ff->gs_selected = sqrtf((ns_vel * ns_vel) + (ew_vel * ew_vel) + 0.5);
                cur_ff->gs_valid = 1;
                if (cur_ff->gs_selected > 0) {
                    float ground_track = atan2(ew_vel, ns_vel) * 180.0 / M_PI;
                    if (ground_track < 0)
                        ground_track += 360;
                    cur_ff->heading = ground_track;
                    cur_ff->heading_type = HT_GROUND_TRACK;
                    cur_ff->heading_valid = 1;
                }
            }
        }
        break;
    }
    default: {
        if (getbits(me, 14, 14)) {
            f->heading_valid = 1;
            f->heading_raw = (uint16_t)getbits(me, 15, 24); /* x 360 / 1024 */
            f->heading_type = 4;
            if (cur_ff) { /* mode_s.c:851-855 */
                cur_ff->heading_valid = 1;
                cur_ff->heading = getbits(me, 15, 24) * 360.0 / 1024.0;
                cur_ff->heading_type = 4;
            }
        }
        const unsigned airspeed = getbits(me, 26, 35);
        if (airspeed) {
            const unsigned speed = (airspeed - 1) * (f->mesub == 4 ? 4 : 1);
            if (getbits(me, 25, 25)) {
                f->tas_valid = 1;
                f->tas = (uint16_t)speed;
            } else {
                f->ias_valid = 1;
                f->ias = (uint16_t)speed;
            }
        }
        break;
    }
    }
    const unsigned vert_rate = getbits(me, 38, 46);
    if (vert_rate) {
        const int rate = (int)(vert_rate - 1) * (getbits(me, 37, 37) ? -64 : 64);
        if (getbits(me, 36, 36)) {
            f->baro_rate = (int16_t)rate;
            f->baro_rate_valid = 1;
        } else {
            f->geom_rate = (int16_t)rate;
            f->geom_rate_valid = 1;
        }
    }
    const unsigned raw_delta = getbits(me, 50, 56);
    if (raw_delta) {
        f->geom_delta_valid = 1;
        f->geom_delta = (int16_t)((int)(raw_delta - 1) * (getbits(me, 49, 49) ? -25 : 25));
    }
}

static void es_surface(const uint8_t *me, orc_fields *f, int check_imf) /* mode_s.c:902-937 */
{
    f->airground = 1;
    f->cpr_valid = 1;
    f->cpr_type = 0; /* CPR_SURFACE */
    const unsigned movement = getbits(me, 6, 12);
    if (movement > 0 && movement < 125)
        f->movement = (uint8_t)movement;
    if (cur_ff && movement > 0 && movement < 125) { /* mode_s.c:911-916 */
        cur_ff->gs_valid = 1;
        cur_ff->gs_selected = cur_ff->gs_v0 = movement_field_v0(movement);
        cur_ff->gs_v2 = movement_field_v2(movement);
    }
    if (getbits(me, 13, 13)) {
        f->heading_valid = 1;
        f->heading_raw = (uint16_t)getbits(me, 14, 20); /* x 360 / 128 */
        f->heading_type = 5;
        if (cur_ff) { /* mode_s.c:920-924 */
            cur_ff->heading_valid = 1;
            cur_ff->heading = getbits(me, 14, 20) * 360.0 / 128.0;
            cur_ff->heading_type = 5;
        }
    }
    if (check_imf && getbits(me, 21, 21))
        set_imf(f);
    f->cpr_odd = (uint8_t)getbits(me, 22, 22);
    f->cpr_lat = getbits(me, 23, 39);
    f->cpr_lon = getbits(me, 40, 56);
}

static void es_airborne(const uint8_t *me, orc_fields *f, int check_imf) /* mode_s.c:939-1022 */
{
    switch (getbits(me, 6, 7)) {
    case 0:
        f->alert_valid = f->spi_valid = 1;
        f->alert = f->spi = 0;
        break;
    case 1: case 2:
        f->alert_valid = 1;
        f->alert = 1;
        break;
    case 3:
        f->alert_valid = f->spi_valid = 1;
        f->alert = 0;
        f->spi = 1;
        break;
    }
    if (check_imf) {
        if (getbits(me, 8, 8))
            set_imf(f);
    } else {
        f->nic_b_valid = 1;
        f->nic_b = (uint8_t)getbits(me, 8, 8);
    }
    const unsigned AC12Field = getbits(me, 9, 20);
    if (f->metype != 0) {
        f->cpr_lat = getbits(me, 23, 39);
        f->cpr_lon = getbits(me, 40, 56);
        if (AC12Field == 0 && f->cpr_lon == 0 && (f->cpr_lat & 0x0fff) == 0 && f->metype == 15) {
            /* stats cpr_filtered++ in the reference */
        } else {
            f->cpr_valid = 1;
            f->cpr_type = 1; /* CPR_AIRBORNE */
            f->cpr_odd = (uint8_t)getbits(me, 22, 22);
        }
    }
    if (AC12Field && f->airground != 1) {
        const int alt = decode_ac12(AC12Field);
        if (alt != ORC_INVALID_ALTITUDE) {
            if (f->metype == 20 || f->metype == 21 || f->metype == 22) {
                f->altitude_geom = alt;
                f->altitude_geom_unit = 0;
                f->altitude_geom_valid = 1;
            } else {
                f->altitude_baro = alt;
                f->altitude_baro_unit = 0;
                f->altitude_baro_valid = 1;
            }
        }
    }
}

/* ---- ME type 29: decodeESTargetStatus, mode_s.c:1058-1249 ---- */
enum { NAVALT_INVALID, NAVALT_UNKNOWN, NAVALT_AIRCRAFT, NAVALT_MCP, NAVALT_FMS };            /* readsb.h:189-195 */
enum { NM_AUTOPILOT = 1, NM_VNAV = 2, NM_ALT_HOLD = 4, NM_APPROACH = 8, NM_LNAV = 16, NM_TCAS = 32 }; /* readsb.h:180-187 */
enum { SILT_INVALID, SILT_UNKNOWN, SILT_PER_SAMPLE, SILT_PER_HOUR };                         /* readsb.pb-c.h:101-106 */
enum { NAVV_MODES = 1, NAVV_HEADING = 2, NAVV_MCP = 4, NAVV_FMS = 8, NAVV_QNH = 16, NAVV_HEADING_V2 = 32 };
enum { ACCV_NAC_P = 1, ACCV_NIC_BARO = 2, ACCV_NIC_A = 4, ACCV_NIC_C = 8, ACCV_GVA = 16, ACCV_SDA = 32 };

static void es_target_status(const uint8_t *me, orc_fields *f, int check_imf)
{
    f->mesub = (uint8_t)getbits(me, 6, 7);
    if (check_imf && getbits(me, 51, 51))
        set_imf(f);

    if (f->mesub == 0 && getbits(me, 11, 11) == 0) { /* version 1 */
        switch (getbits(me, 8, 9)) {
        case 1: f->nav_altitude_source = NAVALT_MCP; break;
        case 2: f->nav_altitude_source = NAVALT_AIRCRAFT; break;
        case 3: f->nav_altitude_source = NAVALT_FMS; break;
        default: break;
        }
        switch (getbits(me, 14, 15)) {
        case 1:
            f->nav_valid |= NAVV_MODES;
            if (f->nav_altitude_source == NAVALT_FMS)
                f->nav_modes |= NM_VNAV;
            else
                f->nav_modes |= NM_AUTOPILOT;
            break;
        case 2:
            f->nav_valid |= NAVV_MODES;
            if (f->nav_altitude_source == NAVALT_FMS)
                f->nav_modes |= NM_VNAV;
            else if (f->nav_altitude_source == NAVALT_AIRCRAFT)
                f->nav_modes |= NM_ALT_HOLD;
            else
                f->nav_modes |= NM_AUTOPILOT;
            break;
        default: break;
        }
        int alt = -1000 + 100 * (int)getbits(me, 16, 25);
        switch (f->nav_altitude_source) {
        case NAVALT_MCP: f->nav_valid |= NAVV_MCP; f->nav_mcp_altitude = alt; break;
        case NAVALT_FMS: f->nav_valid |= NAVV_FMS; f->nav_fms_altitude = alt; break;
        default: break;
        }
        unsigned h_source = getbits(me, 26, 27);
        if (h_source != 0) {
            f->nav_valid |= NAVV_HEADING;
            f->nav_heading_raw = (uint16_t)getbits(me, 28, 36);
            if (cur_ff) { /* mode_s.c:1130-1131 */
                cur_ff->nav_heading_valid = 1;
                cur_ff->nav_heading = getbits(me, 28, 36);
            }
            f->nav_heading_type = getbits(me, 37, 37) ? HT_GROUND_TRACK : HT_MAGNETIC_OR_TRUE;
        }
        switch (getbits(me, 38, 39)) {
        case 1: case 2:
            f->nav_valid |= NAVV_MODES;
            f->nav_modes |= (h_source == 3) ? NM_LNAV : NM_AUTOPILOT;
            break;
        default: break;
        }
        f->acc_valid |= ACCV_NAC_P;
        f->nac_p = (uint8_t)getbits(me, 40, 43);
        f->acc_valid |= ACCV_NIC_BARO;
        f->nic_baro = (uint8_t)getbits(me, 44, 44);
        f->sil = (uint8_t)getbits(me, 45, 46);
        f->sil_type = SILT_UNKNOWN;
        switch (getbits(me, 52, 53)) {
        case 1: f->nav_valid |= NAVV_MODES; break;
        case 2: case 3: f->nav_valid |= NAVV_MODES; f->nav_modes |= NM_TCAS; break;
        case 0: f->nav_modes |= NM_TCAS; break; /* mode_s.c:1186-1190: set without validating the modes */
        }
        f->emergency_valid = 1;
        f->emergency = (uint8_t)getbits(me, 54, 56);
    } else if (f->mesub == 1) { /* version 2 */
        unsigned is_fms = getbits(me, 9, 9);
        unsigned alt_bits = getbits(me, 10, 20);
        if (alt_bits != 0) {
            if (is_fms) {
                f->nav_valid |= NAVV_FMS;
                f->nav_fms_altitude = (int32_t)((alt_bits - 1) * 32);
            } else {
                f->nav_valid |= NAVV_MCP;
                f->nav_mcp_altitude = (int32_t)((alt_bits - 1) * 32);
            }
        }
        unsigned baro_bits = getbits(me, 21, 29);
        if (baro_bits != 0) {
            f->nav_valid |= NAVV_QNH;
            f->nav_qnh_raw = (uint16_t)baro_bits;
            if (cur_ff) { /* mode_s.c:1211-1212 */
                cur_ff->nav_qnh_valid = 1;
                cur_ff->nav_qnh = 800.0 + (baro_bits - 1) * 0.8;
            }
        }
        if (getbits(me, 30, 30)) {
            f->nav_valid |= NAVV_HEADING | NAVV_HEADING_V2;
            f->nav_heading_raw = (uint16_t)getbits(me, 31, 39);
            if (cur_ff) { /* mode_s.c:1216-1219 */
                cur_ff->nav_heading_valid = 1;
                cur_ff->nav_heading = getbits(me, 31, 39) * 180.0 / 256.0;
            }
            f->nav_heading_type = HT_MAGNETIC_OR_TRUE;
        }
        f->acc_valid |= ACCV_NAC_P;
        f->nac_p = (uint8_t)getbits(me, 40, 43);
        f->acc_valid |= ACCV_NIC_BARO;
        f->nic_baro = (uint8_t)getbits(me, 44, 44);
        f->sil = (uint8_t)getbits(me, 45, 46);
        f->sil_type = SILT_UNKNOWN;
        if (getbits(me, 47, 47)) {
            f->nav_valid |= NAVV_MODES;
            f->nav_modes = (uint8_t)((getbits(me, 48, 48) ? NM_AUTOPILOT : 0) | (getbits(me, 49, 49) ? NM_VNAV : 0) |
                                     (getbits(me, 50, 50) ? NM_ALT_HOLD : 0) | (getbits(me, 52, 52) ? NM_APPROACH : 0) |
                                     (getbits(me, 53, 53) ? NM_TCAS : 0) | (getbits(me, 54, 54) ? NM_LNAV : 0));
        }
    }
}

/* ---- ME type 31: decodeESOperationalStatus, mode_s.c:1251-1370; the bit-fields of
 * modesMessage.opstatus (readsb.h:492-524) are collected in a struct and packed at the end ---- */
static void es_operational_status(const uint8_t *me, orc_fields *f, int check_imf)
{
    struct {
        unsigned valid, version, om_acas_ra, om_ident, om_atc, om_saf, cc_acas, cc_cdti, cc_1090_in, cc_arv, cc_ts, cc_tc,
            cc_uat_in, cc_poa, cc_b2_low, cc_lw_valid, cc_lw, hrd, tah;
    } o;
    memset(&o, 0, sizeof o);
    f->mesub = (uint8_t)getbits(me, 6, 8);
    if (check_imf && getbits(me, 56, 56))
        set_imf(f);
    if (f->mesub == 0 || f->mesub == 1) {
        o.valid = 1;
        o.version = getbits(me, 41, 43);
        switch (o.version) {
        case 0:
            if (f->mesub == 0 && getbits(me, 9, 10) == 0) {
                o.cc_acas = !getbits(me, 12, 12);
                o.cc_cdti = getbits(me, 13, 13);
            }
            break;
        case 1:
            if (getbits(me, 25, 26) == 0) {
                o.om_acas_ra = getbits(me, 27, 27);
                o.om_ident = getbits(me, 28, 28);
                o.om_atc = getbits(me, 29, 29);
            }
            if (f->mesub == 0 && getbits(me, 9, 10) == 0 && getbits(me, 13, 14) == 0) {
                o.cc_acas = !getbits(me, 11, 11);
                o.cc_cdti = getbits(me, 12, 12);
                o.cc_arv = getbits(me, 15, 15);
                o.cc_ts = getbits(me, 16, 16);
                o.cc_tc = getbits(me, 17, 18);
            } else if (f->mesub == 1 && getbits(me, 9, 10) == 0 && getbits(me, 13, 14) == 0) {
                o.cc_poa = getbits(me, 11, 11);
                o.cc_cdti = getbits(me, 12, 12);
                o.cc_b2_low = getbits(me, 15, 15);
                o.cc_lw_valid = 1;
                o.cc_lw = getbits(me, 21, 24);
            }
            f->acc_valid |= ACCV_NIC_A;
            f->nic_a = (uint8_t)getbits(me, 44, 44);
            f->acc_valid |= ACCV_NAC_P;
            f->nac_p = (uint8_t)getbits(me, 45, 48);
            f->sil_type = SILT_UNKNOWN;
            f->sil = (uint8_t)getbits(me, 51, 52);
            o.hrd = getbits(me, 54, 54) ? HT_MAGNETIC : HT_TRUE;
            if (f->mesub == 0) {
                f->acc_valid |= ACCV_NIC_BARO;
                f->nic_baro = (uint8_t)getbits(me, 53, 53);
            } else {
                o.tah = getbits(me, 53, 53) ? o.hrd : HT_GROUND_TRACK;
            }
            break;
        case 2:
            if (getbits(me, 25, 26) == 0) {
                o.om_acas_ra = getbits(me, 27, 27);
                o.om_ident = getbits(me, 28, 28);
                o.om_atc = getbits(me, 29, 29);
                o.om_saf = getbits(me, 30, 30);
                f->acc_valid |= ACCV_SDA;
                f->sda = (uint8_t)getbits(me, 31, 32);
            }
            if (f->mesub == 0 && getbits(me, 9, 10) == 0) {
                o.cc_acas = getbits(me, 11, 11); /* inverted sense versus v0 / v1 */
                o.cc_1090_in = getbits(me, 12, 12);
                o.cc_arv = getbits(me, 15, 15);
                o.cc_ts = getbits(me, 16, 16);
                o.cc_tc = getbits(me, 17, 18);
                o.cc_uat_in = getbits(me, 19, 19);
            } else if (f->mesub == 1 && getbits(me, 9, 10) == 0) {
                o.cc_poa = getbits(me, 11, 11);
                o.cc_1090_in = getbits(me, 12, 12);
                o.cc_b2_low = getbits(me, 15, 15);
                o.cc_uat_in = getbits(me, 16, 16);
                f->nac_v_valid = 1;
                f->nac_v = (uint8_t)getbits(me, 17, 19);
                f->acc_valid |= ACCV_NIC_C;
                f->nic_c = (uint8_t)getbits(me, 20, 20);
                o.cc_lw_valid = 1;
                o.cc_lw = getbits(me, 21, 24);
                f->cc_antenna_offset = (uint8_t)getbits(me, 33, 40);
            }
            f->acc_valid |= ACCV_NIC_A;
            f->nic_a = (uint8_t)getbits(me, 44, 44);
            f->acc_valid |= ACCV_NAC_P;
            f->nac_p = (uint8_t)getbits(me, 45, 48);
            f->sil = (uint8_t)getbits(me, 51, 52);
            f->sil_type = getbits(me, 55, 55) ? SILT_PER_SAMPLE : SILT_PER_HOUR;
            o.hrd = getbits(me, 54, 54) ? HT_MAGNETIC : HT_TRUE;
            if (f->mesub == 0) {
                f->acc_valid |= ACCV_GVA;
                f->gva = (uint8_t)getbits(me, 49, 50);
                f->acc_valid |= ACCV_NIC_BARO;
                f->nic_baro = (uint8_t)getbits(me, 53, 53);
            } else {
                o.tah = getbits(me, 53, 53) ? o.hrd : HT_GROUND_TRACK;
            }
            break;
        default: break;
        }
    }
    f->opstatus = o.valid | o.version << 1 | o.om_acas_ra << 4 | o.om_ident << 5 | o.om_atc << 6 | o.om_saf << 7 |
                  o.cc_acas << 8 | o.cc_cdti << 9 | o.cc_1090_in << 10 | o.cc_arv << 11 | o.cc_ts << 12 | o.cc_tc << 13 |
                  o.cc_uat_in << 15 | o.cc_poa << 16 | o.cc_b2_low << 17 | o.cc_lw_valid << 18 | o.cc_lw << 19 |
                  o.hrd << 23 | o.tah << 26;
}

static void extended_squitter(const orc_message *mm, orc_fields *f) /* mode_s.c:1373-1474 */
{
    const uint8_t *me = mm->msg + 4;
    const unsigned metype = getbits(me, 1, 5);
    int check_imf = 0;
    f->metype = (uint8_t)metype;
    if (mm->msgtype == 18) {
        switch (f->CF) {
        case 0: f->addrtype = AT_ADSB_ICAO_NT; break;
        case 1: f->addrtype = AT_ADSB_OTHER; f->addr |= NON_ICAO; break;
        case 2: f->source = SRC_TISB; f->addrtype = AT_TISB_ICAO; check_imf = 1; break;
        case 3:
            f->source = SRC_TISB;
            f->addrtype = AT_TISB_ICAO;
            if (getbits(me, 1, 1))
                set_imf(f);
            return;
        case 5: f->addrtype = AT_TISB_OTHER; f->source = SRC_TISB; f->addr |= NON_ICAO; break;
        case 6: f->addrtype = AT_ADSR_ICAO; f->source = SRC_ADSR; check_imf = 1; break;
        default: f->addrtype = AT_UNKNOWN; f->addr |= NON_ICAO; return;
        }
    }
    switch (metype) {
    case 1: case 2: case 3: case 4: es_ident(me, f); break;
    case 19: es_velocity(me, f, check_imf); break;
    case 5: case 6: case 7: case 8: es_surface(me, f, check_imf); break;
    case 0: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: case 17: case 18:
    case 20: case 21: case 22: es_airborne(me, f, check_imf); break;
    case 23: /* test message, mode_s.c:1024-1036 */
        f->mesub = (uint8_t)getbits(me, 6, 8);
        if (f->mesub == 7) {
            const unsigned ID13Field = getbits(me, 9, 21);
            if (ID13Field) {
                f->squawk_valid = 1;
                f->squawk = (uint16_t)orc_decode_id13(ID13Field);
            }
        }
        break;
    case 28: /* aircraft status, mode_s.c:1038-1057 */
        f->mesub = (uint8_t)getbits(me, 6, 8);
        if (f->mesub == 1) {
            f->emergency_valid = 1;
            f->emergency = (uint8_t)getbits(me, 9, 11);
            const unsigned ID13Field = getbits(me, 12, 24);
            if (ID13Field) {
                f->squawk_valid = 1;
                f->squawk = (uint16_t)orc_decode_id13(ID13Field);
            }
            if (check_imf && getbits(me, 56, 56))
                set_imf(f);
        }
        break;
    case 29: es_target_status(me, f, check_imf); break;
    case 31: es_operational_status(me, f, check_imf); break;
    default: /* 24, 30 and the rest carry nothing the reference decodes */
        break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Comm-B (comm_b.c): DF20/21 do not name the BDS register their MB field holds, so every known    */
/* layout is scored and the best unambiguous one is decoded.  Restated decoder by decoder, with    */
/* the reference's float arithmetic; `store` fills the fields like the reference's second call.    */
/* ------------------------------------------------------------------------------------------ */
enum { CB_UNKNOWN, CB_AMBIGUOUS, CB_EMPTY_RESPONSE, CB_DATALINK_CAPS, CB_GICB_CAPS, CB_AIRCRAFT_IDENT, CB_ACAS_RA,
       CB_VERTICAL_INTENT, CB_TRACK_TURN, CB_HEADING_SPEED }; /* commb_format_t, readsb.h:166-177 */
enum { CBV_ROLL = 1, CBV_GS = 2, CBV_TRACK_RATE = 4, CBV_MACH = 8 };
#define NAVV_QNH_COMMB 64
static unsigned getbit1(const uint8_t *d, unsigned n) { return getbits(d, n, n); }

static int cb_empty(const uint8_t *msg, orc_fields *f, int store) /* comm_b.c:86-98 */
{
    for (unsigned i = 0; i < 7; ++i)
        if (msg[i] != 0)
            return 0;
    if (store)
        f->commb_format = CB_EMPTY_RESPONSE;
    return 56;
}

static int cb_bds10(const uint8_t *msg, orc_fields *f, int store) /* :102-122 */
{
    if (msg[0] != 0x10)
        return 0;
    if (getbits(msg, 10, 14) != 0)
        return 0;
    if (store)
        f->commb_format = CB_DATALINK_CAPS;
    return 56;
}

static int cb_bds17(const uint8_t *msg, orc_fields *f, int store) /* :126-203 */
{
    if (getbits(msg, 25, 56) != 0)
        return 0; /* reserved */
    /* capability bits that are almost never set cost two points each; BDS 2,0 (bit 7) is on almost everything */
    static const unsigned rare[] = {10, 11, 12, 13, 14, 20, 21, 22};
    int score = getbit1(msg, 7) ? 1 : -2;
    for (unsigned i = 0; i < sizeof rare / sizeof rare[0]; ++i)
        score -= 2 * (int)getbit1(msg, rare[i]);
    const unsigned es = getbits(msg, 1, 5); /* the five extended squitter registers come together or not at all */
    if (es == 0x1f)
        score += 5 + (int)getbit1(msg, 6);
    else if (es == 0 && !getbit1(msg, 6))
        score += 1;
    else
        score -= 12;
    const unsigned turn = getbit1(msg, 16), speed = getbit1(msg, 24), intent = getbit1(msg, 9);
    if (turn && speed)
        score += 2 + (int)intent;
    else if (!turn && !speed && !intent)
        score += 1;
    else
        score -= 6;
    if (store)
        f->commb_format = CB_GICB_CAPS;
    return score;
}

static int cb_bds20(const uint8_t *msg, orc_fields *f, int store) /* :207-250 */
{
    if (msg[0] != 0x20)
        return 0;
    char cs[8];
    int score = 8, usable = 1;
    for (unsigned i = 0; i < 8; ++i) {
        cs[i] = ais_charset[getbits(msg, 9 + 6 * i, 14 + 6 * i)];
        const int alnum = (cs[i] >= 'A' && cs[i] <= 'Z') || (cs[i] >= '0' && cs[i] <= '9') || cs[i] == ' ';
        if (alnum)
            score += 6;
        else if (cs[i] == '@')
            usable = 0; /* padding: still a BDS 2,0, but no callsign to take */
        else
            return 0;
    }
    if (store) {
        f->commb_format = CB_AIRCRAFT_IDENT;
        if (usable) {
            memcpy(f->callsign, cs, 8);
            f->callsign_valid = 1;
        }
    }
    return score;
}

static int cb_bds30(const uint8_t *msg, orc_fields *f, int store) /* :254-268 */
{
    if (msg[0] != 0x30)
        return 0;
    if (store)
        f->commb_format = CB_ACAS_RA;
    return 56;
}

/* The shape every value of BDS 4,0 / 5,0 / 6,0 is tested in: a status bit in front of a raw value.
 * Status set (and, where `needs_value`, a non-zero raw value): the decoded value must be plausible and then
 * earns `points`; status clear and nothing in the field (`field_bits`: the raw value, with the sign bit where
 * the reference looks at it): one point; any other combination: not this register.  Returns 0 to give up. */
static int cb_field(unsigned status, unsigned field_bits, unsigned raw, int needs_value, int plausible, int points,
                    int *score)
{
    if (status && (!needs_value || raw != 0)) {
        if (!plausible)
            return 0;
        *score += points;
        return 1;
    }
    if (!status && field_bits == 0) {
        *score += 1;
        return 1;
    }
    return 0;
}

static int cb_bds40(const uint8_t *msg, orc_fields *f, int store) /* :272-434 */
{
    const unsigned mcp_on = getbit1(msg, 1), mcp = getbits(msg, 2, 13), fms_on = getbit1(msg, 14), fms = getbits(msg, 15, 26);
    const unsigned qnh_on = getbit1(msg, 27), qnh = getbits(msg, 28, 39), mode_on = getbit1(msg, 48), mode = getbits(msg, 49, 51);
    const unsigned src_on = getbit1(msg, 54), src = getbits(msg, 55, 56);
    if (!(mcp_on || fms_on || qnh_on || mode_on || src_on))
        return 0;
    const unsigned mcp_alt = mcp * 16, fms_alt = fms * 16; /* feet */
    const float baro_setting = 800 + qnh * 0.1;           /* hPa, as the reference computes it */
    int score = 0;
    if (!cb_field(mcp_on, mcp, mcp, 1, mcp_alt >= 1000 && mcp_alt <= 50000, 13, &score) ||
        !cb_field(fms_on, fms, fms, 1, fms_alt >= 1000 && fms_alt <= 50000, 13, &score) ||
        !cb_field(qnh_on, qnh, qnh, 1, baro_setting >= 900 && baro_setting <= 1100, 13, &score))
        return 0;
    if (getbits(msg, 40, 47) != 0 || getbits(msg, 52, 53) != 0)
        return 0; /* reserved */
    if (!cb_field(mode_on, mode, mode, 0, 1, 4, &score) || !cb_field(src_on, src, src, 0, 1, 3, &score))
        return 0;
    if (mcp_on && fms_on && mcp_alt != fms_alt)
        score -= 4;
    const unsigned alts[2] = {mcp_alt, fms_alt}, on[2] = {mcp_on, fms_on};
    for (int i = 0; i < 2; ++i) /* people select multiples of 500 ft */
        if (on[i] && !(alts[i] % 500 < 16 || alts[i] % 500 > 484))
            score -= 4;
    if (store) {
        static const uint8_t source_of[4] = {NAVALT_UNKNOWN, NAVALT_AIRCRAFT, NAVALT_MCP, NAVALT_FMS};
        f->commb_format = CB_VERTICAL_INTENT;
        if (mcp_on) {
            f->nav_valid |= NAVV_MCP;
            f->nav_mcp_altitude = (int32_t)mcp_alt;
        }
        if (fms_on) {
            f->nav_valid |= NAVV_FMS;
            f->nav_fms_altitude = (int32_t)fms_alt;
        }
        if (qnh_on) {
            f->nav_valid |= NAVV_QNH | NAVV_QNH_COMMB;
            f->nav_qnh_raw = (uint16_t)qnh;
            if (cur_ff) { /* comm_b.c:397-400 */
                cur_ff->nav_qnh_valid = 1;
                cur_ff->nav_qnh = baro_setting;
            }
        }
        if (mode_on) {
            f->nav_valid |= NAVV_MODES;
            f->nav_modes = (uint8_t)(((mode & 4) ? NM_VNAV : 0) | ((mode & 2) ? NM_ALT_HOLD : 0) | ((mode & 1) ? NM_APPROACH : 0));
        }
        f->nav_altitude_source = src_on ? source_of[src] : NAVALT_INVALID;
    }
    return score;
}

static int cb_bds50(const uint8_t *msg, orc_fields *f, int store) /* :438-592 */
{
    const unsigned roll_on = getbit1(msg, 1), roll_neg = getbit1(msg, 2), roll_raw = getbits(msg, 3, 11);
    const unsigned trk_on = getbit1(msg, 12), trk_west = getbit1(msg, 13), trk_raw = getbits(msg, 14, 23);
    const unsigned gs_on = getbit1(msg, 24), gs_raw = getbits(msg, 25, 34);
    const unsigned rate_on = getbit1(msg, 35), rate_neg = getbit1(msg, 36), rate_raw = getbits(msg, 37, 45);
    const unsigned tas_on = getbit1(msg, 46), tas_raw = getbits(msg, 47, 56);
    if (!roll_on || !trk_on || !gs_on || !tas_on)
        return 0;
    float roll = roll_raw * 45.0 / 256.0;
    if (roll_neg)
        roll -= 90.0;
    float track_rate = rate_raw * 8.0 / 256.0;
    if (rate_neg)
        track_rate -= 16;
    float track = trk_raw * 90.0 / 512.0; /* comm_b.c:485-490 */
    if (trk_west)
        track += 180.0;
    const unsigned gs = gs_raw * 2, tas = tas_raw * 2;
    int score = 0;
    if (!cb_field(roll_on, roll_raw | roll_neg, roll_raw, 0, roll >= -40 && roll < 40, 11, &score) ||
        !cb_field(trk_on, trk_raw | trk_west, trk_raw, 0, 1, 12, &score) ||
        !cb_field(gs_on, gs_raw, gs_raw, 1, gs >= 50 && gs <= 700, 11, &score) ||
        !cb_field(rate_on, rate_raw | rate_neg, rate_raw, 0, track_rate >= -10.0 && track_rate <= 10.0, 11, &score) ||
        !cb_field(tas_on, tas_raw, tas_raw, 1, tas >= 50 && tas <= 700, 11, &score))
        return 0;
    /* comm_b.c:542-548 means to compare ground speed and airspeed but subtracts their status bits: |1 - 1| is
     * never more than 150, the penalty never applies */
    if (abs((int)gs_on - (int)tas_on) > 150)
        score -= 6;
    if (rate_on && tas > 0) { /* the turn rate a coordinated turn at this bank and speed has */
        const double turn_rate = 68625 * tan(roll * M_PI / 180.0) / (tas * 20 * M_PI);
        if (fabs(turn_rate - track_rate) > 2.0)
            score -= 6;
    }
    if (store) {
        f->commb_format = CB_TRACK_TURN;
        f->commb_valid |= CBV_ROLL | CBV_GS;
        f->roll_q = (int16_t)((int)roll_raw - (roll_neg ? 512 : 0)); /* roll * 256 / 45 */
        f->heading_valid = 1;
        f->heading_raw = (uint16_t)(trk_raw + (trk_west ? 1024 : 0)); /* (raw * 90 / 512 [+ 180]) * 512 / 90 */
        f->heading_type = HT_GROUND_TRACK;
        f->gs = (uint16_t)gs;
        if (rate_on) {
            f->commb_valid |= CBV_TRACK_RATE;
            f->track_rate_q = (int16_t)((int)rate_raw - (rate_neg ? 512 : 0)); /* rate * 32 */
        }
        f->tas_valid = 1;
        f->tas = (uint16_t)tas;
        if (cur_ff) { /* comm_b.c:561-584 */
            cur_ff->roll_valid = 1;
            cur_ff->roll = roll;
            cur_ff->heading_valid = 1;
            cur_ff->heading = track;
            cur_ff->heading_type = HT_GROUND_TRACK;
            cur_ff->gs_valid = 1;
            cur_ff->gs_v0 = cur_ff->gs_v2 = cur_ff->gs_selected = gs;
            if (rate_on) {
                cur_ff->track_rate_valid = 1;
                cur_ff->track_rate = track_rate;
            }
        }
    }
    return score;
}

static int cb_bds60(const uint8_t *msg, orc_fields *f, int store) /* :596-744 */
{
    const unsigned hdg_on = getbit1(msg, 1), hdg_west = getbit1(msg, 2), hdg_raw = getbits(msg, 3, 12);
    const unsigned ias_on = getbit1(msg, 13), ias = getbits(msg, 14, 23);
    const unsigned mach_on = getbit1(msg, 24), mach_raw = getbits(msg, 25, 34);
    const unsigned baro_on = getbit1(msg, 35), baro_neg = getbit1(msg, 36), baro_raw = getbits(msg, 37, 45);
    const unsigned ins_on = getbit1(msg, 46), ins_neg = getbit1(msg, 47), ins_raw = getbits(msg, 48, 56);
    if (!hdg_on || !ias_on || !mach_on || (!baro_on && !ins_on))
        return 0;
    const float mach = mach_raw * 2.048 / 512;
    float heading = hdg_raw * 90.0 / 512.0; /* comm_b.c:623-628 */
    if (hdg_west)
        heading += 180.0;
    const int baro_rate = (int)baro_raw * 32 - (baro_neg ? 16384 : 0), inertial_rate = (int)ins_raw * 32 - (ins_neg ? 16384 : 0);
    int score = 0;
    if (!cb_field(hdg_on, hdg_raw | hdg_west, hdg_raw, 0, 1, 12, &score) ||
        !cb_field(ias_on, ias, ias, 1, ias >= 50 && ias <= 700, 11, &score) ||
        !cb_field(mach_on, mach_raw, mach_raw, 1, mach >= 0.1 && mach <= 0.9, 11, &score) ||
        !cb_field(baro_on, baro_raw, baro_raw, 0, baro_rate >= -6000 && baro_rate <= 6000, 11, &score) || /* (sign bit not */
        !cb_field(ins_on, ins_raw, ins_raw, 0, inertial_rate >= -6000 && inertial_rate <= 6000, 11, &score)) /* looked at) */
        return 0;
    if (baro_on && ins_on && abs(baro_rate - inertial_rate) > 2000)
        score -= 12;
    if (store) {
        f->commb_format = CB_HEADING_SPEED;
        f->heading_valid = 1;
        f->heading_raw = (uint16_t)(hdg_raw + (hdg_west ? 1024 : 0));
        f->heading_type = HT_MAGNETIC;
        f->ias_valid = 1;
        f->ias = (uint16_t)ias;
        f->commb_valid |= CBV_MACH;
        f->mach_raw = (uint16_t)mach_raw;
        if (cur_ff) { /* comm_b.c:714-728 */
            cur_ff->heading_valid = 1;
            cur_ff->heading = heading;
            cur_ff->heading_type = HT_MAGNETIC;
            cur_ff->mach_valid = 1;
            cur_ff->mach = mach;
        }
        if (baro_on) {
            f->baro_rate_valid = 1;
            f->baro_rate = (int16_t)baro_rate;
        }
        if (ins_on) { /* INS-derived: a "geometric" rate like elsewhere */
            f->geom_rate_valid = 1;
            f->geom_rate = (int16_t)inertial_rate;
        }
    }
    return score;
}

/* decodeCommB, comm_b.c:50-84.  um is what mm->UM holds at that point of decodeModesMessage: the field is
 * only extracted further down (mode_s.c:705 after :669), so the caller passes 0. */
static void comm_b(const uint8_t *mb, orc_fields *f, unsigned dr, unsigned um, unsigned correctedbits)
{
    typedef int (*decoder)(const uint8_t *, orc_fields *, int);
    static const decoder decoders[] = {cb_empty, cb_bds10, cb_bds20, cb_bds30, cb_bds17, cb_bds40, cb_bds50, cb_bds60};
    f->commb_format = CB_UNKNOWN;
    if (dr != 0 || um != 0 || correctedbits > 0)
        return;
    int bestScore = 0, ambiguous = 0;
    decoder bestDecoder = NULL;
    for (unsigned i = 0; i < sizeof decoders / sizeof decoders[0]; ++i) {
        int score = decoders[i](mb, f, 0);
        if (score > bestScore) {
            bestScore = score;
            bestDecoder = decoders[i];
            ambiguous = 0;
        } else if (score == bestScore) {
            ambiguous = 1;
        }
    }
    if (bestDecoder) {
        if (ambiguous)
            f->commb_format = CB_AMBIGUOUS;
        else
            bestDecoder(mb, f, 1);
    }
}

static void fields_mode_s(const orc_message *mm, orc_fields *f) /* mode_s.c:557-715 */
{
    const uint8_t *msg = mm->msg;
    const int t = mm->msgtype;
    memset(f, 0, sizeof *f);
    f->addr = mm->addr;
    f->source = (t == 11) ? SRC_MODE_S_CHECKED : ((t == 17 || t == 18) ? SRC_ADSB : SRC_MODE_S); /* mode_s.c:447-551 */
    if (t == 0 || t == 4 || t == 16 || t == 20) {
        f->AC = (uint16_t)getbits(msg, 20, 32);
        if (f->AC) {
            int unit = 0;
            f->altitude_baro = orc_decode_ac13(f->AC, &unit);
            f->altitude_baro_unit = (uint8_t)unit;
            if (f->altitude_baro != ORC_INVALID_ALTITUDE)
                f->altitude_baro_valid = 1;
        }
    }
    if (t == 11 || t == 17) {
        static const uint8_t ag[8] = {3, 0, 0, 0, 1, 2, 3, 3}; /* CA 0,4,5,6,7 set it; 1-3 leave it alone */
        f->CA = (uint8_t)getbits(msg, 6, 8);
        f->airground = ag[f->CA];
    }
    if (t == 0)
        f->CC = (uint8_t)getbits(msg, 7, 7);
    if (t == 18)
        f->CF = (uint8_t)getbits(msg, 6, 8);
    if (t == 4 || t == 5 || t == 20 || t == 21) {
        f->DR = (uint8_t)getbits(msg, 9, 13);
        f->FS = (uint8_t)getbits(msg, 6, 8);
        f->alert_valid = 1;
        f->spi_valid = 1;
        switch (f->FS) {
        case 0: f->airground = 3; break;
        case 1: f->airground = 1; break;
        case 2: f->airground = 3; f->alert = 1; break;
        case 3: f->airground = 1; f->alert = 1; break;
        case 4: f->airground = 3; f->alert = 1; f->spi = 1; break;
        case 5: f->airground = 3; f->spi = 1; break;
        default: f->spi_valid = 0; f->alert_valid = 0; break;
        }
        f->UM = (uint8_t)getbits(msg, 14, 19);
    }
    if (t == 5 || t == 21) {
        f->ID = (uint16_t)getbits(msg, 20, 32);
        if (f->ID) {
            f->squawk = (uint16_t)orc_decode_id13(f->ID);
            f->squawk_valid = 1;
        }
    }
    if (t >= 24 && t <= 31) {
        f->KE = (uint8_t)getbits(msg, 4, 4);
        f->ND = (uint8_t)getbits(msg, 5, 8);
    }
    if (t == 0 || t == 16) {
        f->RI = (uint8_t)getbits(msg, 14, 17);
        f->SL = (uint8_t)getbits(msg, 9, 11);
        f->VS = (uint8_t)getbits(msg, 6, 6);
        f->airground = f->VS ? 1 : 3;
    }
    if (t == 17 || t == 18)
        extended_squitter(mm, f);
    if (t == 20 || t == 21) /* MB, mode_s.c:666-670 */
        comm_b(msg + 4, f, f->DR, 0, mm->correctedbits);
}

/* decodeModeAMessage (mode_ac.c:168-202) on the record demodulate2400AC reuses within a buffer */
static void fields_mode_ac(orc_fields *mm, unsigned ModeA)
{
    mm->source = 1;   /* SOURCE_MODE_AC */
    mm->addrtype = 8; /* ADDR_MODE_A */
    mm->addr = (ModeA & 0x0000FF7F) | 0x01000000u;
    mm->squawk = (uint16_t)(ModeA & 0x7777);
    mm->squawk_valid = 1;
    mm->spi = (ModeA & 0x0080) ? 1 : 0;
    mm->spi_valid = 1;
    if (!mm->spi) {
        const int modeC = orc_mode_a_to_mode_c(ModeA);
        if (modeC != ORC_INVALID_ALTITUDE) {
            mm->altitude_baro = modeC * 100;
            mm->altitude_baro_unit = 0;
            mm->altitude_baro_valid = 1;
        }
    }
}

/* the fields of one accepted Mode S message (msgtype 0..31), for known-answer tests */
void orc_fields_of(const orc_message *mm, orc_fields *out)
{
    fields_mode_s(mm, out);
}

/* the float-valued members for one accepted Mode S message, straight from its bytes */
void orc_fields_float_of(const orc_message *mm, orc_fields_float *out)
{
    orc_fields scratch;
    memset(out, 0, sizeof *out);
    cur_ff = out;
    fields_mode_s(mm, &scratch);
    cur_ff = NULL;
}

void orc_set_fields_out(orc_ctx *ctx, orc_fields *fields, size_t cap)
{
    ctx->fields_out = fields;
    ctx->fields_cap = cap;
}

static void emit(orc_ctx *ctx, orc_message *out, size_t cap, size_t *nout, const orc_message *mm)
{
    if (*nout < cap)
        out[*nout] = *mm;
    if (ctx->fields_out && *nout < ctx->fields_cap) {
        if (mm->msgtype == 32) {
            fields_mode_ac(&ctx->ac_mm, ((unsigned)mm->msg[0] << 8) | mm->msg[1]);
            ctx->fields_out[*nout] = ctx->ac_mm;
        } else {
            fields_mode_s(mm, &ctx->fields_out[*nout]);
        }
    }
    ++*nout;
}

/* demod_2400.c:236-428 */
static void demod_mode_s(orc_ctx *ctx, const uint16_t *m, unsigned valid_length,
                         uint64_t sample_ts, uint64_t sys_ts, double mean_power, orc_message *out,
                         size_t cap, size_t *nout)
{
    uint32_t mlen = valid_length - ORC_OVERLAP;
    uint64_t sum_scaled_signal_power = 0;
    struct trial_state ts;
    memset(&ts, 0, sizeof ts);

    ctx->ifile_now = sys_ts; /* demod_2400.c:252-255 */

    for (uint32_t j = 0; j < mlen; j++) {
        const uint16_t *pa = &m[j];

        if (!(pa[1] > pa[7] && pa[12] > pa[14] && pa[12] > pa[15]))
            continue; /* demod_2400.c:276 */

        int32_t base_noise = pa[5] + pa[8] + pa[16] + pa[17] + pa[18];
        /* demod_2400.c:285-290: "reduce number of preamble detections if we recently dropped samples" -- the threshold is
         * max(PREAMBLE_THRESHOLD_PIZERO, Modes.preambleThreshold) while the 15-minute statistics hold dropped samples.  Never
         * the case for ifile input; a live feed's host says so through orc_set_recently_dropped (the product's host calls
         * msd_set_preamble_threshold with the same maximum: the statistics window is the host program's) */
        const int thr = ctx->recently_dropped && ctx->threshold < 75 ? 75 : ctx->threshold;
        int32_t ref_level = (int32_t)((uint32_t)base_noise * (uint32_t)thr);
        ref_level >>= 5;

        ts.best = NULL;
        ts.bestscore = -42;
        ts.bestphase = -1;

        int32_t diff_2_3 = pa[2] - pa[3];
        int32_t sum_1_4 = pa[1] + pa[4];
        int32_t diff_10_11 = pa[10] - pa[11];
        int32_t common3456 = sum_1_4 - diff_2_3 + pa[9] + pa[12];

        if (common3456 - diff_10_11 >= ref_level) {
            try_phase(ctx, m, j, 4, &ts);
            try_phase(ctx, m, j, 5, &ts);
        }
        if (common3456 + diff_10_11 >= ref_level) {
            try_phase(ctx, m, j, 6, &ts);
            try_phase(ctx, m, j, 7, &ts);
        }
        if (sum_1_4 + 2 * diff_2_3 + diff_10_11 + pa[12] >= ref_level)
            try_phase(ctx, m, j, 8, &ts);

        if (ts.bestscore == -42)
            continue;

        ctx->st.demod_preambles++;

        if (ts.bestscore < 0) {
            if (ts.bestscore == -1)
                ctx->st.demod_rejected_unknown_icao++;
            else
                ctx->st.demod_rejected_bad++;
            continue;
        }

        int msglen = bits_for_df(ts.best[0] >> 3);

        orc_message mm;
        memset(&mm, 0, sizeof mm);
        mm.timestampMsg = sample_ts + j * 5 + (8 + 56) * 12 + (unsigned)ts.bestphase;
        mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u; /* util.c:79-81 */
        ctx->ifile_now = mm.sysTimestampMsg; /* demod_2400.c:363-366 */
        mm.score = ts.bestscore;
        mm.bestphase = (uint8_t)ts.bestphase;

        int result = decode_accept(ctx, &mm, ts.best);
        if (result < 0) {
            if (result == -1)
                ctx->st.demod_rejected_unknown_icao++;
            else
                ctx->st.demod_rejected_bad++;
            continue;
        }
        ctx->st.demod_accepted[mm.correctedbits]++;
        ctx->st.demod_bestPhase[ts.bestphase - 4]++;

        { /* demod_2400.c:386-408 */
            uint64_t scaled = 0;
            int signal_len = msglen * 12 / 5;
            for (int k = 0; k < signal_len; ++k) {
                uint32_t v = m[j + 19 + k];
                scaled += v * v;
            }
            double signal_power = scaled / 65535.0 / 65535.0;
            mm.signalLevel = signal_power / signal_len;
            ctx->st.signal_power_sum += signal_power;
            ctx->st.signal_power_count += (uint64_t)signal_len;
            sum_scaled_signal_power += scaled;
            if (mm.signalLevel > ctx->st.peak_signal_power)
                ctx->st.peak_signal_power = mm.signalLevel;
            if (mm.signalLevel > 0.50119)
                ctx->st.strong_signal_count++;
        }

        j += (uint32_t)(msglen * 12 / 5); /* demod_2400.c:416 */
        emit(ctx, out, cap, nout, &mm);    /* useModesMessage, demod_2400.c:419 */
    }

    { /* demod_2400.c:422-427 */
        double sum_signal_power = sum_scaled_signal_power / 65535.0 / 65535.0;
        ctx->st.noise_power_sum += (mean_power * mlen - sum_signal_power);
        ctx->st.noise_power_count += mlen;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Mode A/C demodulator (demod_2400.c:522-708)                                                */
/* ------------------------------------------------------------------------------------------ */

static void demod_mode_ac(orc_ctx *ctx, const uint16_t *m, unsigned valid_length,
                          uint64_t sample_ts, uint64_t sys_ts, double mean_level,
                          double mean_power, orc_message *out, size_t cap, size_t *nout)
{
    uint32_t mlen = valid_length - ORC_OVERLAP;
    double noise_stddev = sqrt(mean_power - mean_level * mean_level);
    unsigned noise_level = (unsigned)((mean_power + noise_stddev) * 65535 + 0.5);
    memset(&ctx->ac_mm, 0, sizeof ctx->ac_mm); /* demod_2400.c:523-528: cleared once per buffer */

    for (unsigned f1_sample = 1; f1_sample < mlen; ++f1_sample) {
        if (!(m[f1_sample - 1] < m[f1_sample + 0]))
            continue;
        if (m[f1_sample + 2] > m[f1_sample + 0] || m[f1_sample + 2] > m[f1_sample + 1])
            continue;
        unsigned f1_level = (m[f1_sample + 0] + m[f1_sample + 1]) / 2;
        if (noise_level * 2 > f1_level)
            continue;

        float f1a_power = (float)m[f1_sample] * m[f1_sample];
        float f1b_power = (float)m[f1_sample + 1] * m[f1_sample + 1];
        float fraction = f1b_power / (f1a_power + f1b_power);
        unsigned f1_clock = (unsigned)(25 * (f1_sample + fraction * fraction) + 0.5);

        unsigned f2_clock = f1_clock + (87 * 14);
        unsigned f2_sample = f2_clock / 25;

        if (!(m[f2_sample - 1] < m[f2_sample + 0]))
            continue;
        if (m[f2_sample + 2] > m[f2_sample + 0] || m[f2_sample + 2] > m[f2_sample + 1])
            continue;
        unsigned f2_level = (m[f2_sample + 0] + m[f2_sample + 1]) / 2;
        if (noise_level * 2 > f2_level)
            continue;

        unsigned f1f2_level = (f1_level > f2_level ? f1_level : f2_level);
        float midpoint = sqrtf(noise_level * f1f2_level); /* u32 product, may wrap */
        unsigned signal_threshold = (unsigned)(midpoint * M_SQRT2 + 0.5);
        unsigned noise_threshold = (unsigned)(midpoint / M_SQRT2 + 0.5);

        unsigned uncertain_bits = 0, noisy_bits = 0, bits = 0;
        unsigned clock = f1_clock;
        for (unsigned bit = 0; bit < 20; ++bit, clock += 87) {
            unsigned sample = clock / 25;
            bits <<= 1;
            noisy_bits <<= 1;
            uncertain_bits <<= 1;
            if (m[sample + 2] >= signal_threshold)
                noisy_bits |= 1;
            if (m[sample + 0] >= signal_threshold || m[sample + 1] >= signal_threshold)
                bits |= 1;
            else if (m[sample + 0] > noise_threshold && m[sample + 1] > noise_threshold)
                uncertain_bits |= 1;
        }

        if ((bits & 0x80020) != 0x80020)
            continue;
        if ((bits & 0x0101B) != 0)
            continue;
        if (noisy_bits || uncertain_bits)
            continue;

        /* demod_2400.c:672-685: 00 A4 A2 A1  00 B4 B2 B1  SPI C4 C2 C1  00 D4 D2 D1 */
        static const struct { unsigned from, to; } perm[13] = {
            {0x40000, 0x0010}, {0x20000, 0x1000}, {0x10000, 0x0020}, {0x08000, 0x2000},
            {0x04000, 0x0040}, {0x02000, 0x4000}, {0x00800, 0x0100}, {0x00400, 0x0001},
            {0x00200, 0x0200}, {0x00100, 0x0002}, {0x00080, 0x0400}, {0x00040, 0x0004},
            {0x00004, 0x0080},
        };
        unsigned modeac = 0;
        for (int k = 0; k < 13; ++k)
            if (bits & perm[k].from)
                modeac |= perm[k].to;

        orc_message mm;
        memset(&mm, 0, sizeof mm);
        mm.timestampMsg = sample_ts + f2_clock / 5;
        mm.sysTimestampMsg = sys_ts + (mm.timestampMsg - sample_ts) / 12000u;
        /* mode_ac.c:168-202, identity part */
        mm.msgtype = 32;
        mm.msgbits = 16;
        mm.msg[0] = (uint8_t)(modeac >> 8);
        mm.msg[1] = (uint8_t)modeac;
        mm.addr = (modeac & 0x0000FF7Fu) | 0x01000000u; /* MODES_NON_ICAO_ADDRESS, readsb.h */
        emit(ctx, out, cap, nout, &mm);

        f1_sample += (20 * 87 / 25);
        ctx->st.demod_modeac++;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* buffer and capture level (readsb.c:820-855, sdr_ifile.c:164-237, fifo.c:166-201)           */
/* ------------------------------------------------------------------------------------------ */

void orc_demod_buffer(orc_ctx *ctx, const uint16_t *data, unsigned valid_length,
                      uint64_t sample_ts, uint64_t sys_ts, double mean_level, double mean_power,
                      orc_message *out, size_t cap, size_t *nout)
{
    demod_mode_s(ctx, data, valid_length, sample_ts, sys_ts, mean_power, out, cap, nout);
    if (ctx->mode_ac)
        demod_mode_ac(ctx, data, valid_length, sample_ts, sys_ts, mean_level, mean_power, out, cap,
                      nout);
    ctx->st.samples_processed += valid_length; /* readsb.c:835 */
    ctx->st.buffers++;
    filter_expire(ctx); /* backgroundTasks, readsb.c:331; order fixed as "demod, then expire" */
}

uint64_t orc_replay(orc_ctx *ctx, const void *iq, uint64_t nsamples, orc_message *out, size_t cap,
                    size_t *nout, double *chunk_means, size_t means_cap)
{
    const unsigned bps = (ctx->format == ORC_FMT_UC8) ? 2u : 4u; /* sdr_ifile.c:130-141 */
    const uint8_t *src = iq;
    uint64_t sample_counter = 0, nbuf = 0;
    int eof = 0;
    *nout = 0;

    while (!eof) {
        uint64_t sample_ts = (uint64_t)(sample_counter * 12e6 / 2400000.0); /* sdr_ifile.c:187 */
        uint64_t sys_ts = sample_ts / 12000u;                               /* startup_time = 0 */

        uint64_t left = nsamples - sample_counter;
        unsigned got = (left >= ORC_CHUNK_SAMPLES) ? ORC_CHUNK_SAMPLES : (unsigned)left;
        if (got < ORC_CHUNK_SAMPLES)
            eof = 1; /* short read => EOF (sdr_ifile.c:197-209); an exact multiple therefore
                        yields one more, empty, buffer */

        double mean_level, mean_power;
        orc_convert(ctx, src + sample_counter * bps, &ctx->buf[ORC_OVERLAP], got, &mean_level,
                    &mean_power);
        unsigned valid_length = ORC_OVERLAP + got;

        /* fifo.c:179-188 */
        memcpy(ctx->buf, ctx->carry, sizeof ctx->carry);
        memcpy(ctx->carry, &ctx->buf[valid_length - ORC_OVERLAP], sizeof ctx->carry);

        if (chunk_means && nbuf < means_cap) {
            chunk_means[2 * nbuf] = mean_level;
            chunk_means[2 * nbuf + 1] = mean_power;
        }
        orc_demod_buffer(ctx, ctx->buf, valid_length, sample_ts, sys_ts, mean_level, mean_power,
                         out, cap, nout);
        sample_counter += got;
        ++nbuf;
    }
    return nbuf;
}

/* ------------------------------------------------------------------------------------------ */

orc_ctx *orc_create(int format, int preamble_threshold, int nfix_crc, int mode_ac)
{
    if (format < ORC_FMT_UC8 || format > ORC_FMT_SC16Q11 || nfix_crc < 0 || nfix_crc > 2)
        return NULL; /* MODES_MAX_BITERRORS, crc.h:30 */
    orc_ctx *ctx = calloc(1, sizeof *ctx);
    if (!ctx)
        return NULL;
    ctx->buf = calloc(ORC_CHUNK_SAMPLES + ORC_OVERLAP, sizeof ctx->buf[0]);
    if (!ctx->buf) {
        free(ctx);
        return NULL;
    }
    ctx->format = format;
    ctx->threshold = preamble_threshold;
    ctx->nfix = nfix_crc;
    ctx->mode_ac = mode_ac;
    build_uc8_table();
    build_crc_tables();
    if (nfix_crc >= 1) /* crc.c:367-379 */
        syndrome_tables(nfix_crc, &ctx->tab56, &ctx->ntab56, &ctx->tab112, &ctx->ntab112);
    memset(ctx->filt, 0xFF, sizeof ctx->filt); /* icao_filter.c:67-71 */
    ctx->active = 0;
    ctx->next_flip = 0;
    return ctx;
}

void orc_destroy(orc_ctx *ctx)
{
    if (!ctx)
        return;
    free(ctx->buf);
    free(ctx->q11_table);
    free(ctx);
}

void orc_get_stats(const orc_ctx *ctx, orc_stats *st)
{
    *st = ctx->st;
}

/* ---------------------------------------------------------------------------------------- */
/* wire formats (net_io.c:769-835, 870-896)                                                 */
/* ---------------------------------------------------------------------------------------- */

size_t orc_avr_line(const orc_message *mm, int mlat, char *out)
{
    /* net_io.c:877-893: "@%012" PRIX64 when --mlat and a timestamp exists, else '*'; two upper-case
     * hex digits per byte (printHexDigit, :856-860); ";\n" */
    int n = 0;
    if (mlat && mm->timestampMsg)
        n += sprintf(out + n, "@%012llX", (unsigned long long)mm->timestampMsg);
    else
        out[n++] = '*';
    for (int j = 0; j < mm->msgbits / 8; ++j)
        n += sprintf(out + n, "%02X", mm->msg[j]);
    out[n++] = ';';
    out[n++] = '\n';
    out[n] = 0;
    return (size_t)n;
}

static uint8_t *beast_put(uint8_t *p, uint8_t ch) /* net_io.c:794-797: a data byte 0x1A is sent twice */
{
    *p++ = ch;
    if (ch == 0x1A)
        *p++ = ch;
    return p;
}

size_t orc_beast_frame(const orc_message *mm, uint8_t *out)
{
    const int msgLen = mm->msgbits / 8;
    uint8_t *p = out;
    *p++ = 0x1a;
    if (msgLen == 7)
        *p++ = '2';
    else if (msgLen == 14)
        *p++ = '3';
    else if (msgLen == 2)
        *p++ = '1';
    else
        return 0;
    p = beast_put(p, (uint8_t)(mm->timestampMsg >> 40));
    p = beast_put(p, (uint8_t)(mm->timestampMsg >> 32));
    p = beast_put(p, (uint8_t)(mm->timestampMsg >> 24));
    p = beast_put(p, (uint8_t)(mm->timestampMsg >> 16));
    p = beast_put(p, (uint8_t)(mm->timestampMsg >> 8));
    p = beast_put(p, (uint8_t)(mm->timestampMsg));
    int sig = (int)round(sqrt(mm->signalLevel) * 255); /* net_io.c:819 */
    if (mm->signalLevel > 0 && sig < 1)
        sig = 1;
    if (sig > 255)
        sig = 255;
    p = beast_put(p, (uint8_t)sig);
    for (int j = 0; j < msgLen; ++j)
        p = beast_put(p, mm->msg[j]);
    return (size_t)(p - out);
}
