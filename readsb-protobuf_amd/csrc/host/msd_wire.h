/* msd_wire.h -- the two wire formats readsb forwards accepted messages in:
 * AVR "raw" text lines (modesSendRawOutput, net_io.c:870-896; displayModesMessage --raw,
 * mode_s.c:1786-1798) and Beast binary frames (modesSendBeastOutput, net_io.c:769-835), the rule that decides which
 * messages they carry (modesQueueOutput, net_io.c:1263-1290) and the readers of both (net_io.c:1486-1627, 1656-1764,
 * 2504-2569).  Plain C over msd_message; the caller owns the buffers. */
#ifndef MSD_WIRE_H
#define MSD_WIRE_H

#include <stddef.h>
#include <stdint.h>

#include "modes_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_AVR_MAX 48   /* '@' + 12 hex + 28 hex + ';' + '\n' + NUL */
#define MSD_BEAST_MAX 44 /* 2 + 2 * (7 + 14): every byte after the type may need a 0x1A escape */

/* "*<hex>;\n", or "@<12 hex digits of the 12 MHz timestamp><hex>;\n" when mlat is set and the
 * timestamp is not zero (net_io.c:877-883).  Upper-case hex.  Returns the length (no NUL counted). */
size_t msd_avr_line(const msd_message *mm, int mlat, char *out);

/* 0x1A, type '1' (Mode A/C, 2 bytes) / '2' (56 bit) / '3' (112 bit), 6-byte big-endian timestamp,
 * signal byte round(sqrt(signalLevel) * 255) clamped to 1..255 for a non-zero level, payload; every
 * 0x1A after the type byte is doubled.  Returns the length, 0 for a message it cannot carry. */
size_t msd_beast_frame(const msd_message *mm, uint8_t *out);

/* ---- what modesQueueOutput forwards (net_io.c:1263-1290) ---- */
/* The raw and the Beast output carry a message iff `net_verbatim || correctedbits < 2`: a message that needed two bit
 * repairs (--aggressive) only goes out with --net-verbatim.  (Neither is an mlat result here: SOURCE_MLAT never comes
 * out of the demodulator.) */
int msd_wire_forwards(const msd_message *mm, int net_verbatim);
/* mm->verbatim (mode_s.c:427-429): the bytes as they were received, before modesChecksumFix -- what both outputs send
 * with --net-verbatim (net_io.c:775,874).  msd_message carries the repaired bytes, the syndrome of the received ones
 * (crc) and the number of repaired bits; the repaired bits are the one or two positions in [5, msgbits) whose
 * single-bit syndromes xor to that syndrome (crc.c:184-354 keeps a syndrome only while exactly one such pattern has it,
 * so the pattern is found again by search).  Returns the number of bits put back (0..2), -1 if no pattern fits (not a
 * record of this library); out receives msgbits / 8 bytes either way. */
int msd_wire_verbatim(const msd_message *mm, uint8_t out[14]);
/* The two writers with the forwarding rule and --net-verbatim applied: 0 = not forwarded, nothing written. */
size_t msd_avr_line_out(const msd_message *mm, int mlat, int net_verbatim, char *out);
size_t msd_beast_frame_out(const msd_message *mm, int net_verbatim, uint8_t *out);

/* ---- the readers: the inverse framing (net_io.c:1486-1627 decodeBinMessage behind the READ_MODE_BEAST scanner of
 * net_io.c:2504-2569; decodeHexMessage net_io.c:1656-1764).  They produce msd_message records for msd_decode_fields /
 * msd_decode_fields_device: timestampMsg and signalLevel = (byte / 255)^2 from the frame, msgtype = DF (32: Mode A/C),
 * crc = modesChecksum of the bytes, addr = the AA field (DF 11, 17, 18) or the checksum (address/parity formats;
 * mode_s.c:559-562), iid for DF 11; score, correctedbits, bestphase, sysTimestampMsg are zero -- the sender's
 * acceptance is not repeated. ---- */
typedef struct msd_beast_reader {
    uint8_t buf[256]; /* an incomplete frame kept between calls (the longest escaped frame is 44 bytes) */
    size_t len;
    int mode_ac;             /* deliver type '1' frames (Modes.mode_ac; otherwise they only count, net_io.c:1500-1508) */
    uint64_t frames;         /* messages delivered */
    uint64_t modeac_ignored; /* type '1' frames with mode_ac off */
    uint64_t other_frames;   /* well-formed frames of the types this reader has no use for ('4', '5', 'H') */
    uint64_t garbage_bytes;  /* bytes skipped in front of a 0x1A or behind an unknown type */
} msd_beast_reader;
void msd_beast_reader_init(msd_beast_reader *r, int mode_ac);
/* Appends n bytes of the stream; every complete frame goes to fn in order.  Returns the messages delivered by this call. */
size_t msd_beast_reader_feed(msd_beast_reader *r, const uint8_t *data, size_t n, msd_message_fn fn, void *user);
/* One line of the AVR family ("*hex;", ":hex;", "@<12 hex>hex;", "%<12 hex>hex;", "<<12 hex><2 hex signal>hex;", surrounding
 * white space allowed): 1 and *out filled, or 0 for anything decodeHexMessage drops (no ';', unknown prefix, wrong
 * length, a non-hex digit, a Mode A/C length with mode_ac off).  The timestamp digits are skipped like the
 * reference does (net_io.c:1700-1706) -- timestampMsg stays 0 -- except that '@' / '%' / '<' lines keep theirs in
 * out->timestampMsg for the caller that wants it (`keep_timestamp`). */
int msd_avr_parse_line(const char *line, int mode_ac, int keep_timestamp, msd_message *out);

#ifdef __cplusplus
}
#endif
#endif
