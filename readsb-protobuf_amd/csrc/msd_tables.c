/*
 * msd_tables.c -- the small constant tables the kernels keep in LDS, computed once on the host.
 * Compile with -ffp-contract=off: the UC8 table must round exactly like the reference's x86-64
 * build (float products and sum rounded separately, correctly rounded sqrtf).
 */
#include "msd_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The reference's UC8 magnitude for an (I, Q) byte pair (convert.c:35-61).  Its table slot is the
 * little-endian u16 of the pair, I + 256*Q, filled by a loop whose outer variable is that slot's
 * high byte -- so the loop's "i" is Q.  fI is evaluated in double and rounded to float once. */
static uint16_t uc8_magnitude(int ibyte, int qbyte)
{
    float a = (qbyte - 127.5) / 127.5;
    float b = (ibyte - 127.5) / 127.5;
    float magsq = a * a + b * b;
    if (magsq > 1)
        magsq = 1;
    float mag = sqrtf(magsq);
    return (uint16_t)(mag * 65535.0f + 0.5f);
}

static int fold(int byte)
{
    /* (b - 127.5) is +-(k + 0.5) with k = b-128 for b >= 128 and 127-b below; squaring drops
     * the sign, so the magnitude only depends on k. */
    return byte >= 128 ? byte - 128 : 127 - byte;
}

uint32_t msd_crc24(const msd_tables *t, const uint8_t *msg, int nbits)
{
    /* crc.c:67-82: table-driven remainder over all but the last three bytes, which are xored in */
    int n = nbits / 8;
    uint32_t rem = 0;
    for (int i = 0; i < n - 3; ++i)
        rem = ((rem << 8) ^ t->crc_byte[msg[i] ^ (rem >> 16)]) & 0xffffffu;
    return rem ^ ((uint32_t)msg[n - 3] << 16) ^ ((uint32_t)msg[n - 2] << 8) ^ msg[n - 1];
}

static void build_slicer_tables(msd_tables *t);

/* msd_tables.synhash: the first odd multiplier from the golden-ratio constant on that puts no more than four of a
 * table's syndromes into one bucket (a handful of tries: 107 keys in 64 buckets, 51 in 32). */
static void build_syndrome_hash(msd_tables *t)
{
    memset(t->synhash, 0, sizeof t->synhash);
    for (int k = 0; k < 2; ++k) {
        const uint32_t *src = k ? t->syn112 : t->syn56;
        const uint32_t n = k ? t->nsyn112 : t->nsyn56, lg = k ? MSD_SYNH_LG112 : MSD_SYNH_LG56;
        uint32_t *dst = t->synhash + (k ? 4u << MSD_SYNH_LG56 : 0u);
        uint32_t mul = 0x9E3779B1u;
        for (;; mul += 2) {
            uint8_t fill[1u << MSD_SYNH_LG112] = {0};
            int ok = 1;
            for (uint32_t i = 0; i < n && ok; ++i)
                ok = ++fill[((src[i] & 0xffffffu) * mul) >> (32 - lg)] <= 4;
            if (ok)
                break;
        }
        t->synhash_mul[k] = mul;
        uint8_t fill[1u << MSD_SYNH_LG112] = {0};
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t b = ((src[i] & 0xffffffu) * mul) >> (32 - lg);
            dst[4 * b + fill[b]++] = src[i];
        }
    }
}

static int cmp_u24(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a & 0xffffffu, y = *(const uint32_t *)b & 0xffffffu;
    return (x > y) - (x < y);
}

void msd_tables_build(msd_tables *t, int nfix_crc)
{
    memset(t, 0, sizeof *t);
    for (int q = 0; q < 256; ++q)
        for (int i = 0; i < 256; ++i)
            t->uc8_full[i + 256 * q] = uc8_magnitude(i, q);
    for (int kq = 0; kq < 128; ++kq)
        for (int ki = 0; ki < 128; ++ki)
            t->uc8_folded[kq * MSD_LUT_STRIDE + ki] = uc8_magnitude(128 + ki, 128 + kq);
    for (int kq = 0; kq < 128; ++kq)
        for (int ki = 0; ki < 128; ++ki)
            t->uc8_scan[MSD_LUT_SCAN_INDEX(kq, ki)] = uc8_magnitude(128 + ki, 128 + kq);

    for (uint32_t b = 0; b < 256; ++b) { /* crc.c:46-57, generator 0xfff409 (crc.c:31) */
        uint32_t c = b << 16;
        for (int k = 0; k < 8; ++k)
            c = (c & 0x800000u) ? ((c << 1) ^ 0xfff409u) : (c << 1);
        t->crc_byte[b] = c & 0xffffffu;
    }

    build_slicer_tables(t);

    if (nfix_crc >= 1) {
        /* crc.c:367-372 with max_correct = max_detect = 1: one entry per bit 5..bits-1 holding the
         * syndrome of that single-bit error, sorted by syndrome.  Packed as syndrome | bit << 24. */
        uint8_t probe[14];
        for (int bits = 56; bits <= 112; bits += 56) {
            uint32_t *tab = (bits == 56) ? t->syn56 : t->syn112;
            uint32_t n = 0;
            for (int i = 5; i < bits; ++i) {
                memset(probe, 0, sizeof probe);
                probe[i >> 3] = (uint8_t)(0x80u >> (i & 7));
                tab[n++] = msd_crc24(t, probe, bits) | ((uint32_t)i << 24);
            }
            qsort(tab, n, sizeof tab[0], cmp_u24);
            if (bits == 56)
                t->nsyn56 = n;
            else
                t->nsyn112 = n;
        }
    }
    build_syndrome_hash(t);
}

/* The slicer / CRC tables of the scan kernel (layout: MSD_SL_* in msd_internal.h). */
static void build_slicer_tables(msd_tables *t)
{
    uint8_t *perm = (uint8_t *)&t->slicer[MSD_SL_PERM];
    for (int q = 0; q < 5; ++q) {
        /* bit k of group 0 of trial phase 4 + q: t = 95 + (4 + q) + 12 k twelfths behind pa[0] */
        uint32_t qoff = 0;
        int bit_of[5];
        for (int k = 0; k < 5; ++k) {
            const int tt = 99 + q + 12 * k, c = tt % 5, sample = tt / 5;
            bit_of[c] = k;
            qoff |= (uint32_t)(2 * sample) << (6 * c);
        }
        t->slicer[MSD_SL_QOFF + q] = qoff;
        for (int x = 0; x < 32; ++x) {
            int v = 0;
            for (int c = 0; c < 5; ++c)
                if ((x >> (4 - c)) & 1)
                    v |= 1 << (4 - bit_of[c]);
            perm[q * 32 + x] = (uint8_t)v;
        }
    }
    for (int bits = 56; bits <= 112; bits += 56) {
        const uint32_t base = bits == 112 ? MSD_SL_GLONG : MSD_SL_GSHORT;
        const uint32_t rows = bits == 112 ? MSD_SL_GLONG_ROWS : MSD_SL_GSHORT_ROWS;
        for (uint32_t g = 0; g < rows; ++g)
            for (int v = 0; v < 32; ++v) {
                uint8_t msg[14];
                memset(msg, 0, sizeof msg);
                for (int k = 0; k < 5; ++k) {
                    const int n = 5 * (int)g + k;
                    if (n < bits && ((v >> (4 - k)) & 1))
                        msg[n >> 3] |= (uint8_t)(0x80u >> (n & 7));
                }
                t->slicer[base + 32 * g + (uint32_t)v] = msd_crc24(t, msg, bits);
            }
    }
}

/* --aggressive (Modes.nfix_crc == 2): prepareErrorTable(bits, 2, 4), crc.c:184-354,374-379, as an
 * open-addressing hash table for the device.  An error pattern of one or two bits out of bits
 * 5..bits-1 is correctable iff no other pattern of up to two bits (crc.c:236-258) and no pattern of three
 * or four bits (flagCollisions, crc.c:155-178,266-287) has the same syndrome.  Instead of the
 * reference's sorted table and one binary search per four-bit pattern, a 16 Mi-entry census over
 * all 24-bit syndromes is taken (low bits: patterns of <= 2 bits seen, saturating; bit 7: a 3- or
 * 4-bit pattern maps here), which makes the 5.4 M patterns of a 112-bit message a few milliseconds.
 * Entry: syndrome | errors << 24 | bit[0] << 32 | bit[1] << 40 (0xff: none); vacant = all ones;
 * slot = MSD_FIX2_HASH(syndrome, log2_slots), linear probing. */
uint64_t *msd_fix2_table(const msd_tables *t, int bits, uint32_t *log2_slots)
{
    const int lo = 5, n = bits - lo;
    uint32_t single[112];
    uint8_t probe[14];
    for (int i = lo; i < bits; ++i) {
        memset(probe, 0, sizeof probe);
        probe[i >> 3] = (uint8_t)(0x80u >> (i & 7));
        single[i - lo] = msd_crc24(t, probe, bits);
    }
    uint8_t *census = calloc(1u << 24, 1);
    if (!census)
        return NULL;
    for (int a = 0; a < n; ++a) {
        const uint32_t sa = single[a];
        if ((census[sa] & 3) < 2)
            census[sa]++;
        for (int b = a + 1; b < n; ++b) {
            const uint32_t sb = sa ^ single[b];
            if ((census[sb] & 3) < 2)
                census[sb]++;
            for (int c = b + 1; c < n; ++c) {
                const uint32_t sc = sb ^ single[c];
                census[sc] |= 0x80;
                for (int d = c + 1; d < n; ++d)
                    census[sc ^ single[d]] |= 0x80;
            }
        }
    }
    const uint32_t lg = bits == 56 ? 12 : 14; /* 1326 / 5778 patterns at most: load factor <= 0.36 */
    uint64_t *tab = malloc(sizeof(uint64_t) << lg);
    if (!tab) {
        free(census);
        return NULL;
    }
    memset(tab, 0xff, sizeof(uint64_t) << lg);
    for (int a = 0; a < n; ++a)
        for (int b = a; b < n; ++b) { /* b == a: the single-bit pattern */
            const uint32_t syn = b == a ? single[a] : single[a] ^ single[b];
            if (census[syn] != 1)
                continue;
            uint64_t e = syn | ((uint64_t)(b == a ? 1 : 2) << 24) | ((uint64_t)(a + lo) << 32) |
                         ((uint64_t)(b == a ? 0xff : b + lo) << 40);
            uint32_t h = MSD_FIX2_HASH(syn, lg);
            while (tab[h] != ~0ull)
                h = (h + 1) & ((1u << lg) - 1);
            tab[h] = e;
        }
    free(census);
    *log2_slots = lg;
    return tab;
}

/* modesChecksumDiagnose against the --aggressive tables on the host (tests; the kernels do the same
 * probe): number of wrong bits (0 for a zero syndrome), -1 if uncorrectable. */
int msd_fix2_diagnose(int bits, uint32_t syndrome, int bit[2])
{
    static uint64_t *tab[2];
    static uint32_t lg[2];
    const int k = bits == 112;
    bit[0] = bit[1] = -1;
    if (syndrome == 0)
        return 0;
    if (!tab[k]) {
        msd_tables *t = malloc(sizeof *t);
        if (!t)
            return -1;
        msd_tables_build(t, 0);
        tab[k] = msd_fix2_table(t, bits, &lg[k]);
        free(t);
        if (!tab[k])
            return -1;
    }
    for (uint32_t h = MSD_FIX2_HASH(syndrome, lg[k]);; h = (h + 1) & ((1u << lg[k]) - 1)) {
        const uint64_t e = tab[k][h];
        if (e == ~0ull)
            return -1;
        if ((e & 0xffffffu) == syndrome) {
            bit[0] = (int)((e >> 32) & 0xff);
            bit[1] = ((e >> 40) & 0xff) == 0xff ? -1 : (int)((e >> 40) & 0xff);
            return (int)((e >> 24) & 0xff);
        }
    }
}

/* init_sc16q11_lookup, convert.c:271-295: entry ((i >> lose) << bits) | (q >> lose) for i, q = 0, 2^lose, ... < 2048 */
void msd_sc16q11_table_build(int bits, uint16_t *out)
{
    const int lose = 11 - bits;
    for (int i = 0; i < 2048; i += 1 << lose)
        for (int q = 0; q < 2048; q += 1 << lose) {
            const float fI = (float)(i / 2048.0), fQ = (float)(q / 2048.0);
            const float sq_i = fI * fI, sq_q = fQ * fQ;
            float magsq = sq_i + sq_q;
            if (magsq > 1)
                magsq = 1;
            const float mag = sqrtf(magsq);
            out[((unsigned)(i >> lose) << bits) | (unsigned)(q >> lose)] = (uint16_t)(mag * 65535.0f + 0.5f);
        }
}

/* Checks the folding identity the kernels rely on; returns the number of mismatching slots. */
int msd_tables_selftest(const msd_tables *t)
{
    int bad = 0;
    for (int q = 0; q < 256; ++q)
        for (int i = 0; i < 256; ++i)
            if (t->uc8_full[i + 256 * q] != t->uc8_folded[fold(q) * MSD_LUT_STRIDE + fold(i)] ||
                t->uc8_full[i + 256 * q] != t->uc8_scan[MSD_LUT_SCAN_INDEX(fold(q), fold(i))])
                ++bad;
    /* every single-bit syndrome is found in its bucket, with its bit */
    for (int k = 0; k < 2; ++k) {
        const uint32_t *src = k ? t->syn112 : t->syn56;
        const uint32_t n = k ? t->nsyn112 : t->nsyn56, lg = k ? MSD_SYNH_LG112 : MSD_SYNH_LG56;
        const uint32_t *tab = t->synhash + (k ? 4u << MSD_SYNH_LG56 : 0u);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t b = ((src[i] & 0xffffffu) * t->synhash_mul[k]) >> (32 - lg);
            int found = 0;
            for (int j = 0; j < 4; ++j)
                found += tab[4 * b + j] == src[i];
            if (found != 1)
                ++bad;
        }
    }
    /* slicer tables against the closed form of demod_2400.c:98-177: bit n of trial phase 4 + q is
     * correlator t % 5 at sample pa + t / 5 with t = 95 + (4 + q) + 12 n */
    const uint8_t *perm = (const uint8_t *)&t->slicer[MSD_SL_PERM];
    for (int q = 0; q < 5; ++q)
        for (int n = 0; n < 112; ++n) {
            const int g = n / 5, k = n % 5, tt = 99 + q + 12 * n, c = tt % 5;
            const int sample = 12 * g + (int)((t->slicer[MSD_SL_QOFF + q] >> (6 * c)) & 63u) / 2;
            if (sample != tt / 5 || perm[q * 32 + (1 << (4 - c))] != (1 << (4 - k)))
                ++bad;
        }
    /* per-group syndromes against modesChecksum on pseudo-random messages */
    uint32_t x = 0x2545F491u;
    for (int trial = 0; trial < 2000; ++trial) {
        uint8_t msg[14];
        const int bits = (trial & 1) ? 112 : 56;
        memset(msg, 0, sizeof msg);
        for (int i = 0; i < bits / 8; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            msg[i] = (uint8_t)(x >> 11);
        }
        uint32_t syn = 0;
        for (int g = 0; 5 * g < bits; ++g) {
            int v = 0;
            for (int k = 0; k < 5; ++k) {
                const int n = 5 * g + k;
                if (n < bits && (msg[n >> 3] & (0x80u >> (n & 7))))
                    v |= 1 << (4 - k);
            }
            syn ^= t->slicer[(bits == 112 ? MSD_SL_GLONG : MSD_SL_GSHORT) + 32 * (uint32_t)g + (uint32_t)v];
        }
        if (syn != msd_crc24(t, msg, bits))
            ++bad;
    }
    return bad;
}
