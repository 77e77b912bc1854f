/* msd_wire.c -- AVR raw lines and Beast binary frames of accepted messages (see msd_wire.h). */
#include "msd_wire.h"

#include <math.h>

static const char HEX[] = "0123456789ABCDEF";

size_t msd_avr_line(const msd_message *mm, int mlat, char *out)
{
    char *p = out;
    const int nbytes = mm->msgbits / 8;
    if (mlat && mm->timestampMsg) { /* net_io.c:877-881: the 48-bit counter, big-endian, 12 digits */
        *p++ = '@';
        for (int shift = 44; shift >= 0; shift -= 4)
            *p++ = HEX[(mm->timestampMsg >> shift) & 0xf];
    } else {
        *p++ = '*';
    }
    for (int j = 0; j < nbytes; ++j) {
        *p++ = HEX[mm->msg[j] >> 4];
        *p++ = HEX[mm->msg[j] & 0xf];
    }
    *p++ = ';';
    *p++ = '\n';
    *p = 0;
    return (size_t)(p - out);
}

size_t msd_beast_frame(const msd_message *mm, uint8_t *out)
{
    const int nbytes = mm->msgbits / 8;
    uint8_t body[6 + 1 + 14];
    uint8_t type;
    if (nbytes == 7)
        type = '2';
    else if (nbytes == 14)
        type = '3';
    else if (nbytes == 2)
        type = '1';
    else
        return 0; /* net_io.c:789-791 */
    for (int i = 0; i < 6; ++i)
        body[i] = (uint8_t)(mm->timestampMsg >> (40 - 8 * i));
    int sig = (int)round(sqrt(mm->signalLevel) * 255); /* net_io.c:819-823 */
    if (mm->signalLevel > 0 && sig < 1)
        sig = 1;
    if (sig > 255)
        sig = 255;
    body[6] = (uint8_t)sig;
    for (int j = 0; j < nbytes; ++j)
        body[7 + j] = mm->msg[j];
    uint8_t *p = out;
    *p++ = 0x1a;
    *p++ = type;
    for (int i = 0; i < 7 + nbytes; ++i) {
        *p++ = body[i];
        if (body[i] == 0x1a)
            *p++ = 0x1a;
    }
    return (size_t)(p - out);
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* modesChecksum (crc.c:31,67-82): remainder of the first n - 3 bytes under the generator 0xfff409, xor the last three */
static uint32_t wire_crc24(const uint8_t *msg, int nbytes)
{
    uint32_t rem = 0;
    for (int i = 0; i < nbytes - 3; ++i) {
        rem ^= (uint32_t)msg[i] << 16;
        for (int b = 0; b < 8; ++b)
            rem = (rem & 0x800000u) ? ((rem << 1) ^ 0xfff409u) & 0xffffffu : (rem << 1) & 0xffffffu;
    }
    return rem ^ ((uint32_t)msg[nbytes - 3] << 16) ^ ((uint32_t)msg[nbytes - 2] << 8) ^ (uint32_t)msg[nbytes - 1];
}

int msd_wire_forwards(const msd_message *mm, int net_verbatim)
{
    return net_verbatim || mm->correctedbits < 2; /* net_io.c:1272-1276, :1278-1285 */
}

int msd_wire_verbatim(const msd_message *mm, uint8_t out[14])
{
    const int nbytes = mm->msgbits / 8, nbits = mm->msgbits;
    for (int j = 0; j < nbytes && j < 14; ++j)
        out[j] = mm->msg[j];
    if (mm->correctedbits == 0 || (nbits != 56 && nbits != 112))
        return 0;
    /* the syndrome modesChecksumDiagnose was asked about: DF 11 masks the interrogator id out (mode_s.c:476-480) */
    const uint32_t want = mm->msgtype == 11 ? (mm->crc & 0xffff80u) : mm->crc;
    uint32_t syn[112];
    uint8_t probe[14];
    for (int i = 5; i < nbits; ++i) { /* crc.c:216: the first five bits (the DF) are never repaired */
        for (int j = 0; j < 14; ++j)
            probe[j] = 0;
        probe[i >> 3] = (uint8_t)(0x80u >> (i & 7));
        syn[i] = wire_crc24(probe, nbytes);
    }
    int b0 = -1, b1 = -1;
    if (mm->correctedbits == 1) {
        for (int i = 5; i < nbits && b0 < 0; ++i)
            if (syn[i] == want)
                b0 = i;
    } else {
        for (int i = 5; i < nbits && b0 < 0; ++i)
            for (int k = i + 1; k < nbits; ++k)
                if ((syn[i] ^ syn[k]) == want) {
                    b0 = i;
                    b1 = k;
                    break;
                }
    }
    if (b0 < 0)
        return -1;
    out[b0 >> 3] ^= (uint8_t)(0x80u >> (b0 & 7)); /* modesChecksumFix is its own inverse (crc.c:417-425) */
    if (b1 >= 0)
        out[b1 >> 3] ^= (uint8_t)(0x80u >> (b1 & 7));
    return b1 >= 0 ? 2 : 1;
}

static const msd_message *wire_outgoing(const msd_message *mm, int net_verbatim, msd_message *tmp)
{
    if (!msd_wire_forwards(mm, net_verbatim))
        return NULL;
    if (!net_verbatim || mm->correctedbits == 0)
        return mm;
    *tmp = *mm;
    msd_wire_verbatim(mm, tmp->msg); /* net_io.c:775,874: msg = Modes.net_verbatim ? mm->verbatim : mm->msg */
    return tmp;
}

size_t msd_avr_line_out(const msd_message *mm, int mlat, int net_verbatim, char *out)
{
    msd_message tmp;
    const msd_message *o = wire_outgoing(mm, net_verbatim, &tmp);
    return o ? msd_avr_line(o, mlat, out) : 0;
}

size_t msd_beast_frame_out(const msd_message *mm, int net_verbatim, uint8_t *out)
{
    msd_message tmp;
    const msd_message *o = wire_outgoing(mm, net_verbatim, &tmp);
    return o ? msd_beast_frame(o, out) : 0;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* what every reader fills in once it has the bytes */
static void wire_message(msd_message *mm, const uint8_t *bytes, int nbytes, uint64_t timestamp, double level)
{
    uint8_t *raw = (uint8_t *)mm;
    for (size_t i = 0; i < sizeof *mm; ++i)
        raw[i] = 0;
    mm->timestampMsg = timestamp;
    mm->signalLevel = level;
    for (int j = 0; j < nbytes; ++j)
        mm->msg[j] = bytes[j];
    mm->msgbits = (uint8_t)(8 * nbytes);
    if (nbytes == 2) { /* decodeModeAMessage, mode_ac.c:168-202 */
        const uint32_t modeac = ((uint32_t)bytes[0] << 8) | bytes[1];
        mm->msgtype = 32;
        mm->addr = (modeac & 0x0000FF7Fu) | (1u << 24);
        return;
    }
    const unsigned df = bytes[0] >> 3;
    mm->msgtype = (uint8_t)df;
    mm->crc = wire_crc24(bytes, nbytes);
    if (df == 11 || df == 17 || df == 18) { /* mode_s.c:559-562: AA */
        mm->addr = ((uint32_t)bytes[1] << 16) | ((uint32_t)bytes[2] << 8) | bytes[3];
        if (df == 11)
            mm->iid = (uint8_t)(mm->crc & 0x7fu);
    } else {
        mm->addr = mm->crc; /* address/parity: the checksum is the address */
    }
}

void msd_beast_reader_init(msd_beast_reader *r, int mode_ac)
{
    r->len = 0;
    r->mode_ac = mode_ac;
    r->frames = r->modeac_ignored = r->other_frames = r->garbage_bytes = 0;
}

/* One pass of the READ_MODE_BEAST scanner (net_io.c:2504-2569) over buf[0..len): returns the bytes consumed; an
 * incomplete frame at the end is left for the next call. */
static size_t beast_scan(msd_beast_reader *r, const uint8_t *buf, size_t len, msd_message_fn fn, void *user, size_t *delivered)
{
    size_t som = 0;
    while (som < len) {
        size_t p = som;
        while (p < len && buf[p] != 0x1a)
            ++p;
        r->garbage_bytes += p - som;
        if (p == len)
            return len; /* no frame start in what is left */
        som = p;
        if (som + 1 >= len)
            break; /* the type byte has not arrived yet */
        const uint8_t type = buf[som + 1];
        size_t body; /* unescaped bytes behind the type byte */
        if (type == '1')
            body = 2 + 7;
        else if (type == '2')
            body = 7 + 7;
        else if (type == '3' || type == '4' || type == '5')
            body = 14 + 7;
        else if (type == 'H') { /* GNS HULC: 0x1A 'H' id len payload */
            if (som + 3 >= len)
                break;
            if (buf[som + 3] > 24) {
                ++som;
                ++r->garbage_bytes;
                continue;
            }
            body = (size_t)buf[som + 3] + 2;
        } else { /* not a frame: skip this 0x1A and look again (net_io.c:2541-2544) */
            ++som;
            ++r->garbage_bytes;
            continue;
        }
        /* the end of the frame, doubled 0x1A bytes counted (net_io.c:2547-2552) */
        uint8_t plain[64];
        size_t q = som + 2, got = 0;
        while (got < body && q < len) {
            const uint8_t ch = buf[q++];
            plain[got++] = ch;
            if (ch == 0x1a) {
                if (q >= len) { /* the second half of the pair is still to come */
                    --got;
                    --q;
                    break;
                }
                ++q;
            }
        }
        if (got < body)
            break; /* incomplete: retry when more has arrived */
        if (type == '1' || type == '2' || type == '3') {
            if (type == '1' && !r->mode_ac) {
                ++r->modeac_ignored;
            } else {
                uint64_t ts = 0;
                for (int j = 0; j < 6; ++j)
                    ts = (ts << 8) | plain[j];
                const double lvl = plain[6] / 255.0; /* net_io.c:1563-1565 */
                msd_message mm;
                wire_message(&mm, plain + 7, (int)body - 7, ts, lvl * lvl);
                ++r->frames;
                ++*delivered;
                if (fn)
                    fn(&mm, user);
            }
        } else {
            ++r->other_frames;
        }
        som = q;
    }
    return som;
}

size_t msd_beast_reader_feed(msd_beast_reader *r, const uint8_t *data, size_t n, msd_message_fn fn, void *user)
{
    size_t delivered = 0;
    while (n) {
        /* what is pending plus as much of the new data as fits: what a scan leaves behind is less than one frame (at most
         * 2 + 2 * 26 bytes), so there is always room for more */
        size_t take = sizeof r->buf - r->len;
        if (take > n)
            take = n;
        for (size_t i = 0; i < take; ++i)
            r->buf[r->len + i] = data[i];
        r->len += take;
        data += take;
        n -= take;
        const size_t used = beast_scan(r, r->buf, r->len, fn, user, &delivered);
        for (size_t i = used; i < r->len; ++i)
            r->buf[i - used] = r->buf[i];
        r->len -= used;
    }
    return delivered;
}

static int hexval(int c)
{
    if (c >= '0' && c <= '9')
        return c - '0';
    if (c >= 'A' && c <= 'F')
        return c - 'A' + 10;
    if (c >= 'a' && c <= 'f')
        return c - 'a' + 10;
    return -1;
}

int msd_avr_parse_line(const char *line, int mode_ac, int keep_timestamp, msd_message *out)
{
    const char *hex = line;
    size_t l = 0;
    while (hex[l])
        ++l;
    while (l && (hex[l - 1] == ' ' || (hex[l - 1] >= '\t' && hex[l - 1] <= '\r')))
        --l;
    while (l && (*hex == ' ' || (*hex >= '\t' && *hex <= '\r'))) {
        ++hex;
        --l;
    }
    if (!l || hex[l - 1] != ';')
        return 0; /* not complete */
    double level = 0.0;
    uint64_t ts = 0;
    size_t skip;
    switch (hex[0]) {
    case '<': skip = 15; break; /* '<' + 12 timestamp digits + 2 signal digits */
    case '@':
    case '%': skip = 13; break;
    case '*':
    case ':': skip = 1; break;
    default: return 0;
    }
    if (l < skip + 1)
        return 0;
    if (skip > 1) {
        for (size_t i = 1; i < 13; ++i) {
            const int v = hexval(hex[i]);
            if (v < 0) { /* (the reference does not look at these digits at all) */
                ts = 0;
                break;
            }
            ts = (ts << 4) | (uint64_t)v;
        }
        if (skip == 15) {
            const int hi = hexval(hex[13]), lo = hexval(hex[14]);
            level = (double)((hi << 4) | lo) / 255.0; /* net_io.c:1690-1691, whatever the digits are */
            level *= level;
        }
    }
    hex += skip;
    l -= skip + 1;
    if (l != 4 && l != 14 && l != 28)
        return 0;
    if (l == 4 && !mode_ac)
        return 0;
    uint8_t bytes[14];
    for (size_t j = 0; j < l; j += 2) {
        const int hi = hexval(hex[j]), lo = hexval(hex[j + 1]);
        if (hi < 0 || lo < 0)
            return 0;
        bytes[j / 2] = (uint8_t)((hi << 4) | lo);
    }
    wire_message(out, bytes, (int)(l / 2), keep_timestamp ? ts : 0, level);
    return 1;
}
