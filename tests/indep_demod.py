"""A second reading of the reference's receive path, in numpy / plain Python -- TEST INFRASTRUCTURE, and on purpose not
derived from oracle/modes_oracle.c: it was written from the reference's sources alone (the file:line citations below),
by somebody who had not looked at the oracle, so that `oracle == this` on the same capture is a statement about two
readings of demod_2400.c / mode_s.c / crc.c / icao_filter.c / convert.c and not about one reading copied three times.
(The reference itself cannot be built in this image -- readsb.h:86 needs protobuf-c -- so this is the closest thing to a
second witness there is.)  Slow (Python loop over preamble hits): meant for captures of a few buffers.

Scope: ifile replay of a UC8 / SC16 / SC16Q11 capture through convert_*_nodc or (--dcfilter) convert_*_generic, demodulate2400, scoreModesMessage,
decodeModesMessage's CRC / address / filter part, modesChecksum + single-bit repair (--fix, the default) or none
(--no-fix), the ICAO filter with its two tables, and demodulate2400AC + decodeModeAMessage's acceptance -- the ordered
message list and the demodulator counters of stats.h:61-80, for --no-fix, --fix and --aggressive."""
import numpy as np

BUF = 131072           # MODES_MAG_BUF_SAMPLES (readsb.h)
OVERLAP = 326          # Modes.trailing_samples = (8 + 112 + 16) * 1e-6 * 2.4e6, readsb.c:198
POLY = 0xFFF409        # MODES_GENERATOR_POLY (crc.c)

# ---------------------------------------------------------------------------------------------- convert.c
def uc8_table():
    """convert.c:35-61: float arithmetic on (i - 127.5) / 127.5, clamp, sqrtf, (uint16_t)(mag * 65535.0f + 0.5f)."""
    v = ((np.arange(256, dtype=np.float64) - 127.5) / 127.5).astype(np.float32)
    magsq = (v[:, None] * v[:, None]).astype(np.float32) + (v[None, :] * v[None, :]).astype(np.float32)
    magsq = np.minimum(magsq.astype(np.float32), np.float32(1.0))
    mag = np.sqrt(magsq, dtype=np.float32)
    return (mag * np.float32(65535.0) + np.float32(0.5)).astype(np.float32).astype(np.uint16)  # [hi byte][lo byte], symmetric


def dc_block(f, sample_rate=2.4e6):
    """convert.c:135-139 (and :187-191, :396-400) with the state of init_converter (:478-482): a one-pole DC estimate
    per channel, z = fI * dc_a + z * dc_b in float arithmetic, one sample after the other through the whole stream."""
    import math
    dc_b = np.float32(math.exp(-2.0 * math.pi * 1.0 / sample_rate))
    dc_a = np.float32(1.0 - float(dc_b))
    out = np.empty_like(f)
    for ch in range(2):
        a = (f[:, ch] * dc_a).astype(np.float32)
        z = np.float32(0.0)
        zs = np.empty(len(a), dtype=np.float32)
        for i in range(len(a)):
            z = a[i] + z * dc_b  # np.float32 scalars: two roundings, as in C without contraction
            zs[i] = z
        out[:, ch] = f[:, ch] - zs
    return out


def convert(fmt, raw, dc=False, sample_rate=2.4e6):
    """-> (uint16 magnitudes, per-sample level terms, per-sample power terms, float sums?); the terms are what the
    converter sums (convert.c:63-111 integers for UC8 without --dcfilter; floats for every other converter)."""
    if fmt == "uc8" and not dc:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 2)
        mag = uc8_table()[b[:, 1], b[:, 0]]  # uc8_lookup[le16 of the pair]: high byte = second byte
        return mag, mag.astype(np.uint64), mag.astype(np.uint64) * mag.astype(np.uint64), False
    if fmt == "uc8":  # convert_uc8_generic, convert.c:113-162
        s = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 2).astype(np.float32)
        f = ((s - np.float32(127.5)) / np.float32(127.5)).astype(np.float32)
    else:
        s = np.frombuffer(raw, dtype="<i2").reshape(-1, 2).astype(np.float32)
        f = s / np.float32(32768.0 if fmt == "sc16" else 2048.0)
    if dc:
        f = dc_block(f, sample_rate)
    magsq = (f[:, 0] * f[:, 0]).astype(np.float32) + (f[:, 1] * f[:, 1]).astype(np.float32)
    magsq = np.minimum(magsq.astype(np.float32), np.float32(1.0))
    magf = np.sqrt(magsq, dtype=np.float32)
    mag = (magf * np.float32(65535.0) + np.float32(0.5)).astype(np.float32).astype(np.uint16)
    return mag, magf, magsq, True


def buffer_means(float_sums, level_terms, power_terms):
    n = len(level_terms)
    if n == 0:
        return float("nan"), float("nan")  # 0 / 0 in every converter: the empty buffer behind a capture that ends on a buffer boundary
    if not float_sums:
        sl, sp = int(level_terms.sum(dtype=np.uint64)), int(power_terms.sum(dtype=np.uint64))
        return sl / 65536.0 / n, sp / 65535.0 / 65535.0 / n
    # float accumulators, one addition per sample, in order
    sl = np.add.accumulate(level_terms, dtype=np.float32)[-1]
    sp = np.add.accumulate(power_terms, dtype=np.float32)[-1]
    return float(np.float32(sl) / np.float32(n)), float(np.float32(sp) / np.float32(n))


# ---------------------------------------------------------------------------------------------- crc.c
def _crc_table():
    t = []
    for i in range(256):
        c = i << 16
        for _ in range(8):
            c = ((c << 1) ^ POLY) if (c & 0x800000) else (c << 1)
        t.append(c & 0xFFFFFF)
    return t


CRC_TABLE = _crc_table()


def checksum(msg, bits):
    """modesChecksum, crc.c:67-82"""
    n = bits // 8
    rem = 0
    for i in range(n - 3):
        rem = ((rem << 8) ^ CRC_TABLE[msg[i] ^ ((rem & 0xFF0000) >> 16)]) & 0xFFFFFF
    return rem ^ (msg[n - 3] << 16) ^ (msg[n - 2] << 8) ^ msg[n - 1]


def _single_bit_syndromes():
    out = []
    for i in range(112):
        msg = bytearray(14)
        msg[i // 8] ^= 1 << (7 - (i & 7))
        out.append(checksum(msg, 112))
    return out


SINGLE = _single_bit_syndromes()


_tables = {}


def error_table(bits, nfix):
    """prepareErrorTable, crc.c:184-350, as modesChecksumInit calls it (:353-383): syndrome -> tuple of wrong bits.
    --fix: (bits, 1, 1) -- every single wrong bit but the DF's five (:214).  --aggressive: (bits, 2, 4) -- one and two
    wrong bits; every syndrome that two of those patterns share is dropped altogether (:231-251), and so is every one
    that a pattern of three or four wrong bits also produces (flagCollisions, :150-177, :254-283)."""
    if nfix == 0:
        return None
    if (bits, nfix) in _tables:
        return _tables[(bits, nfix)]
    off = 112 - bits
    pos = np.arange(5, bits)
    s1 = np.array([SINGLE[i + off] for i in pos], dtype=np.int64)
    table = {int(s): (int(i),) for s, i in zip(s1, pos)}
    if nfix >= 2:
        ii, jj = np.triu_indices(len(pos), k=1)        # i < j
        s2 = s1[ii] ^ s1[jj]
        every = np.concatenate([s1, s2])
        uniq, counts = np.unique(every, return_counts=True)
        shared = set(uniq[counts > 1].tolist())
        table = {s: b for s, b in table.items() if s not in shared}
        for s, i, j in zip(s2.tolist(), pos[ii].tolist(), pos[jj].tolist()):
            if s not in shared:
                table[s] = (i, j)
        keys = np.array(sorted(table), dtype=np.int64)
        flagged = set()
        # three wrong bits i < j < k, four wrong bits i < j < k < l: a pair (i, j) with a single k > j or a pair (k, l), k > j
        for a in range(0, len(s2), 256):
            sa, ja = s2[a: a + 256, None], jj[a: a + 256, None]
            t3 = (sa ^ s1[None, :])[np.arange(len(pos))[None, :] > ja]
            t4 = (sa ^ s2[None, :])[ii[None, :] > ja]
            for t in (t3, t4):
                flagged.update(t[np.isin(t, keys)].tolist())
        table = {s: b for s, b in table.items() if s not in flagged}
    _tables[(bits, nfix)] = table
    return table


# ---------------------------------------------------------------------------------------------- icao_filter.c
class IcaoFilter:
    SIZE = 8192
    EMPTY = 0xFFFFFFFF
    TTL = 60000

    def __init__(self, literal=False):
        self.a = [self.EMPTY] * self.SIZE
        self.b = [self.EMPTY] * self.SIZE
        self.active = self.a
        self.next_flip = 0
        # What the tables hold, as sets: a short cut for test() once a table is crowded (30 000 aircraft fill it, and a
        # miss then walks all 8192 slots).  Sound because entries are only ever wiped table by table and add() stores
        # the second copy only after the first: an address is found on its probe path iff the table holds it.
        # literal=True always walks the table (tests compare the two).
        self.literal = literal
        self.held = {id(self.a): set(), id(self.b): set()}

    @staticmethod
    def hash(a):
        """Jenkins one-at-a-time over three bytes, icao_filter.c:44-65 (32-bit wrap-around)"""
        M = 0xFFFFFFFF
        h = 0
        for byte in (a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF):
            h = (h + byte) & M
            h = (h + (h << 10)) & M
            h ^= h >> 6
        h = (h + (h << 3)) & M
        h ^= h >> 11
        h = (h + (h << 15)) & M
        return h & (IcaoFilter.SIZE - 1)

    def add(self, addr):
        t = self.active
        h0 = h = self.hash(addr)
        while t[h] != self.EMPTY and t[h] != addr:
            h = (h + 1) & (self.SIZE - 1)
            if h == h0:
                return
        if t[h] == self.EMPTY:
            t[h] = addr
            self.held[id(t)].add(addr)
        h0 = h = self.hash(addr & 0xFFFF)
        while t[h] != self.EMPTY and (t[h] & 0xFFFF) != (addr & 0xFFFF):
            h = (h + 1) & (self.SIZE - 1)
            if h == h0:
                return
        if t[h] == self.EMPTY:
            t[h] = addr

    def test(self, addr):
        if not self.literal:
            return addr in self.held[id(self.a)] or addr in self.held[id(self.b)]
        for t in (self.a, self.b):
            h0 = h = self.hash(addr)
            while t[h] != self.EMPTY and t[h] != addr:
                h = (h + 1) & (self.SIZE - 1)
                if h == h0:
                    break
            if t[h] == addr:
                return True
        return False

    def expire(self, now):
        if now >= self.next_flip:
            if self.active is self.a:
                del self.held[id(self.b)]
                self.b = [self.EMPTY] * self.SIZE
                self.held[id(self.b)] = set()
                self.active = self.b
            else:
                del self.held[id(self.a)]
                self.a = [self.EMPTY] * self.SIZE
                self.held[id(self.a)] = set()
                self.active = self.a
            self.next_flip = now + self.TTL


# ---------------------------------------------------------------------------------------------- demod_2400.c
# slice_byte, demod_2400.c:97-177: (correlator, sample offset) of the eight bits of a byte that starts in phase p,
# and how far the pointer moves; the next byte starts in phase (p + 1) % 5.
BYTE_TAPS = {
    0: ([(0, 0), (2, 2), (4, 4), (1, 7), (3, 9), (0, 12), (2, 14), (4, 16)], 19),
    1: ([(1, 0), (3, 2), (0, 5), (2, 7), (4, 9), (1, 12), (3, 14), (0, 17)], 19),
    2: ([(2, 0), (4, 2), (1, 5), (3, 7), (0, 10), (2, 12), (4, 14), (1, 17)], 19),
    3: ([(3, 0), (0, 3), (2, 5), (4, 7), (1, 10), (3, 12), (0, 15), (2, 17)], 19),
    4: ([(4, 0), (1, 3), (3, 5), (0, 8), (2, 10), (4, 12), (1, 15), (3, 17)], 20),
}


def _message_taps(phase):
    corr, off, base = [], [], 0
    for _ in range(14):
        taps, adv = BYTE_TAPS[phase]
        for c, o in taps:
            corr.append(c)
            off.append(base + o)
        base += adv
        phase = (phase + 1) % 5
    return np.array(corr), np.array(off)


MSG_TAPS = {p: _message_taps(p) for p in range(5)}


def correlator_signs(m):
    """slice_phase0..4 > 0 at every sample, demod_2400.c:73-93"""
    x = np.concatenate([m.astype(np.int64), np.zeros(4, dtype=np.int64)])
    m0, m1, m2, m3 = x[:-3], x[1:-2], x[2:-1], x[3:]
    return np.stack([
        18 * m0 - 15 * m1 - 3 * m2 > 0,
        14 * m0 - 5 * m1 - 9 * m2 > 0,
        16 * m0 + 5 * m1 - 20 * m2 > 0,
        7 * m0 + 11 * m1 - 18 * m2 > 0,
        4 * m0 + 15 * m1 - 20 * m2 + m3 > 0,
    ])


def msg_bits_by_type(df):
    return 112 if (df & 0x10) else 56  # modesMessageLenByType, mode_s.c:81-83


class Receiver:
    def __init__(self, fmt="uc8", threshold=58, nfix=1, mode_ac=False, startup_time=0, dc_filter=False, literal_filter=False):
        self.fmt, self.threshold, self.mode_ac, self.dc_filter = fmt, threshold, mode_ac, dc_filter
        self.recently_dropped = False  # Modes.stats_15min.samples_dropped != 0 (demod_2400.c:285-290); the host program's to set
        self.tab56, self.tab112 = error_table(56, nfix), error_table(112, nfix)
        self.filter = IcaoFilter(literal=literal_filter)
        self.startup_time = startup_time
        self.ifile_now = startup_time
        self.stats = dict(demod_preambles=0, demod_rejected_bad=0, demod_rejected_unknown_icao=0, demod_accepted=[0, 0, 0],
                          demod_preamblePhase=[0] * 5, demod_bestPhase=[0] * 5, demod_modeac=0, strong_signal_count=0,
                          noise_power_count=0, signal_power_count=0, noise_power_sum=0.0, signal_power_sum=0.0,
                          peak_signal_power=0.0)
        self.messages = []

    # ---- crc.c:389-412
    def diagnose(self, syndrome, bits):
        """-> None (uncorrectable) or the list of wrong bits"""
        if syndrome == 0:
            return []
        table = self.tab56 if bits == 56 else self.tab112
        if table is None or syndrome not in table:
            return None
        return list(table[syndrome])

    @staticmethod
    def correct_aa(addr, ei):  # mode_s.c:266-281
        for bit in ei:
            if 8 <= bit <= 31:
                addr ^= 1 << (31 - bit)
        return addr

    # ---- mode_s.c:311-409
    def score(self, msg, validbits):
        if validbits < 56:
            return -2
        df = msg[0] >> 3
        bits = msg_bits_by_type(df)
        if validbits < bits:
            return -2
        if not any(msg[: bits // 8]):
            return -2
        crc = checksum(msg, bits)
        if df in (0, 4, 5, 16) or 24 <= df <= 31:
            return 1000 if self.filter.test(crc) else -1
        if df == 11:
            iid = crc & 0x7F
            crc &= 0xFFFF80
            addr = (msg[1] << 16) | (msg[2] << 8) | msg[3]
            ei = self.diagnose(crc, bits)
            if ei is None or len(ei) > 1:
                return -2
            addr = self.correct_aa(addr, ei)
            if iid == 0:
                return (1600 if self.filter.test(addr) else 750) // (len(ei) + 1)
            return 1000 // (len(ei) + 1) if self.filter.test(addr) else -1
        if df in (17, 18):
            ei = self.diagnose(crc, bits)
            if ei is None:
                return -2
            addr = self.correct_aa((msg[1] << 16) | (msg[2] << 8) | msg[3], ei)
            return (1800 if self.filter.test(addr) else 1400) // (len(ei) + 1)
        if df in (20, 21):
            return 1000 if self.filter.test(crc) else -2
        return -2

    # ---- mode_s.c:424-555,560-562,717-726: what decides acceptance and the filter; -> (result, record)
    def decode(self, msg):
        msg = bytearray(msg)
        if not any(msg[:7]):
            return -2, None
        df = msg[0] >> 3
        bits = msg_bits_by_type(df)
        crc = checksum(msg, bits)
        corrected, addr, iid = 0, 0, 0
        if df in (0, 4, 5, 16) or 24 <= df <= 31:
            if not self.filter.test(crc):
                return -1, None
            addr = crc
        elif df == 11:
            iid = crc & 0x7F
            if crc & 0xFFFF80:
                ei = self.diagnose(crc & 0xFFFF80, bits)
                if ei is None or len(ei) > 1:
                    return -2, None
                corrected = len(ei)
                for bit in ei:
                    msg[bit >> 3] ^= 1 << (7 - (bit & 7))
                if not self.filter.test((msg[1] << 16) | (msg[2] << 8) | msg[3]):
                    return -1, None
        elif df in (17, 18):
            if crc != 0:
                ei = self.diagnose(crc, bits)
                if ei is None:
                    return -2, None
                addr1 = (msg[1] << 16) | (msg[2] << 8) | msg[3]
                corrected = len(ei)
                for bit in ei:
                    msg[bit >> 3] ^= 1 << (7 - (bit & 7))
                addr2 = (msg[1] << 16) | (msg[2] << 8) | msg[3]
                if addr1 != addr2 and not self.filter.test(addr2):
                    return -1, None
        elif df in (20, 21):
            if not self.filter.test(crc):
                return -1, None
            addr = crc
        else:
            return -2, None
        if df in (11, 17, 18):
            addr = (msg[1] << 16) | (msg[2] << 8) | msg[3]
        if not corrected and (df == 17 or (df == 11 and iid == 0)):
            self.filter.add(addr)
        return 0, dict(msg=bytes(msg), msgtype=df, msgbits=bits, crc=crc, correctedbits=corrected, addr=addr, iid=iid)

    # ---- demod_2400.c:183-229
    def score_phase(self, try_phase, signs, j, best):
        self.stats["demod_preamblePhase"][try_phase - 4] += 1
        start = j + 19 + try_phase // 5
        corr, off = MSG_TAPS[try_phase % 5]
        df = int(np.packbits(signs[corr[:8], start + off[:8]])[0]) >> 3
        if df in (0, 4, 5, 11):
            nbytes = 7
        elif df in (16, 17, 18, 20, 21, 24):
            nbytes = 14
        else:
            nbytes = 1
        if nbytes > 1:
            msg = bytes(np.packbits(signs[corr[: 8 * nbytes], start + off[: 8 * nbytes]]))
            score = self.score(msg, nbytes * 8)
        else:
            msg, score = None, -2
        if score > best[0]:
            best[0], best[1], best[2] = score, try_phase, msg

    # ---- demod_2400.c:236-428
    def demodulate(self, m, mlen, sample_ts, sys_ts, mean_power):
        st = self.stats
        self.ifile_now = sys_ts
        x = m.astype(np.int64)
        pa = lambda d: x[d: d + mlen]
        pre = (pa(1) > pa(7)) & (pa(12) > pa(14)) & (pa(12) > pa(15))
        base_noise = pa(5) + pa(8) + pa(16) + pa(17) + pa(18)
        ref = (base_noise * (max(75, self.threshold) if self.recently_dropped else self.threshold)) >> 5  # demod_2400.c:285-292
        d23, s14, d1011 = pa(2) - pa(3), pa(1) + pa(4), pa(10) - pa(11)
        common = s14 - d23 + pa(9) + pa(12)
        t0 = pre & (common - d1011 >= ref)
        t1 = pre & (common + d1011 >= ref)
        t2 = pre & (s14 + 2 * d23 + d1011 + pa(12) >= ref)
        signs = correlator_signs(m)
        sum_scaled = 0
        resume = 0
        for j in np.flatnonzero(t0 | t1 | t2):
            j = int(j)
            if j < resume:
                continue
            best = [-42, -1, None]
            if t0[j]:
                self.score_phase(4, signs, j, best)
                self.score_phase(5, signs, j, best)
            if t1[j]:
                self.score_phase(6, signs, j, best)
                self.score_phase(7, signs, j, best)
            if t2[j]:
                self.score_phase(8, signs, j, best)
            bestscore, bestphase, bestmsg = best
            st["demod_preambles"] += 1
            if bestscore < 0:
                st["demod_rejected_unknown_icao" if bestscore == -1 else "demod_rejected_bad"] += 1
                continue
            msglen = msg_bits_by_type(bestmsg[0] >> 3)
            ts = sample_ts + j * 5 + (8 + 56) * 12 + bestphase
            sys_msg = sys_ts + (ts - sample_ts) // 12000
            self.ifile_now = sys_msg
            result, rec = self.decode(bestmsg)
            if result < 0:
                st["demod_rejected_unknown_icao" if result == -1 else "demod_rejected_bad"] += 1
                continue
            st["demod_accepted"][rec["correctedbits"]] += 1
            st["demod_bestPhase"][bestphase - 4] += 1
            signal_len = msglen * 12 // 5
            seg = x[j + 19: j + 19 + signal_len]
            scaled = int((seg * seg).sum())
            signal_power = scaled / 65535.0 / 65535.0
            level = signal_power / signal_len
            st["signal_power_sum"] += signal_power
            st["signal_power_count"] += signal_len
            sum_scaled += scaled
            if level > st["peak_signal_power"]:
                st["peak_signal_power"] = level
            if level > 0.50119:
                st["strong_signal_count"] += 1
            resume = j + signal_len + 1  # j += msglen * 12 / 5, then the loop's ++j
            rec.update(timestampMsg=ts, sysTimestampMsg=sys_msg, score=bestscore, bestphase=bestphase, signalLevel=level)
            self.messages.append(rec)
        st["noise_power_sum"] += mean_power * mlen - sum_scaled / 65535.0 / 65535.0
        st["noise_power_count"] += mlen

    # ---- demod_2400.c:522-708 + mode_ac.c:168-202
    def demodulate_ac(self, m, mlen, sample_ts, sys_ts, mean_level, mean_power):
        import math
        if mlen == 0:
            return  # (its noise level would be a NaN turned into an unsigned; the loop does not run)
        f32 = np.float32
        x = [int(v) for v in m] + [0] * 8
        var = mean_power - mean_level * mean_level
        if var >= 0:
            noise_level = int((mean_power + math.sqrt(var)) * 65535 + 0.5)
        else:
            # sqrt of a negative number (a buffer of one or two samples whose float mean of squares was rounded below the
            # square of the mean): a NaN, and (unsigned) of a NaN is undefined in C.  The reference's x86-64 build turns it
            # into 0 (cvttsd2si gives 0x8000000000000000, the low half is kept) -- which is what is restated here.
            noise_level = 0

        def pulse(s):
            """rising edge, quiet third sample, 6 dB above the noise (:581-594) -> level or None"""
            if not (x[s - 1] < x[s]):
                return None
            if x[s + 2] > x[s] or x[s + 2] > x[s + 1]:
                return None
            level = (x[s] + x[s + 1]) // 2
            if noise_level * 2 > level:
                return None
            return level

        f1_sample = 0
        while True:
            f1_sample += 1  # the for loop's ++f1_sample (it starts at 1)
            if f1_sample >= mlen:
                break
            f1_level = pulse(f1_sample)
            if f1_level is None:
                continue
            f1a = f32(x[f1_sample]) * f32(x[f1_sample])
            f1b = f32(x[f1_sample + 1]) * f32(x[f1_sample + 1])
            fraction = f1b / (f1a + f1b)
            f1_clock = int(float(f32(25) * (f32(f1_sample) + fraction * fraction)) + 0.5)
            f2_clock = f1_clock + 87 * 14
            f2_sample = f2_clock // 25
            f2_level = pulse(f2_sample)
            if f2_level is None:
                continue
            f1f2 = max(f1_level, f2_level)
            midpoint = np.sqrt(f32((noise_level * f1f2) & 0xFFFFFFFF), dtype=f32)
            signal_threshold = int(float(midpoint) * math.sqrt(2.0) + 0.5)
            noise_threshold = int(float(midpoint) / math.sqrt(2.0) + 0.5)
            bits = noisy = uncertain = 0
            clock = f1_clock
            for _ in range(20):
                s = clock // 25
                bits <<= 1
                noisy <<= 1
                uncertain <<= 1
                if x[s + 2] >= signal_threshold:
                    noisy |= 1
                if x[s] >= signal_threshold or x[s + 1] >= signal_threshold:
                    bits |= 1
                elif x[s] > noise_threshold and x[s + 1] > noise_threshold:
                    uncertain |= 1
                clock += 87
            if (bits & 0x80020) != 0x80020 or (bits & 0x0101B) != 0 or noisy or uncertain:
                continue
            modeac = 0
            for src, dst in ((0x40000, 0x0010), (0x20000, 0x1000), (0x10000, 0x0020), (0x08000, 0x2000), (0x04000, 0x0040),
                             (0x02000, 0x4000), (0x00800, 0x0100), (0x00400, 0x0001), (0x00200, 0x0200), (0x00100, 0x0002),
                             (0x00080, 0x0400), (0x00040, 0x0004), (0x00004, 0x0080)):
                if bits & src:
                    modeac |= dst
            ts = sample_ts + f2_clock // 5
            self.messages.append(dict(msg=bytes([modeac >> 8, modeac & 0xFF]), msgtype=32, msgbits=16, correctedbits=0,
                                      addr=(modeac & 0xFF7F) | (1 << 24), timestampMsg=ts,
                                      sysTimestampMsg=sys_ts + (ts - sample_ts) // 12000))
            f1_sample += 20 * 87 // 25
            self.stats["demod_modeac"] += 1

    # ---- sdr_ifile.c:164-237 (lossless, i.e. throttled, feed) + fifo.c:168-188 + readsb.c:820-855,331
    def replay(self, raw):
        bps = 2 if self.fmt == "uc8" else 4
        raw = bytes(raw)
        nsamples = len(raw) // bps
        whole = convert(self.fmt, raw[: nsamples * bps], True) if self.dc_filter else None  # the filter runs through the stream
        tail = np.zeros(OVERLAP, dtype=np.uint16)  # calloc'ed overlap buffer, fifo.c:47
        counter = 0
        while True:
            n = min(BUF, nsamples - counter)
            if whole is not None:
                mag, lvl, pwr, float_sums = (whole[0][counter: counter + n], whole[1][counter: counter + n],
                                             whole[2][counter: counter + n], whole[3])
            else:  # every other converter is a function of the sample alone: one buffer at a time
                mag, lvl, pwr, float_sums = convert(self.fmt, raw[counter * bps: (counter + n) * bps], False)
            data = np.concatenate([tail, mag])
            sample_ts = int(counter * 12e6 / 2.4e6)
            sys_ts = sample_ts // 12000 + self.startup_time
            mean_level, mean_power = buffer_means(float_sums, lvl, pwr)
            tail = data[len(data) - OVERLAP:]
            self.demodulate(data, n, sample_ts, sys_ts, mean_power)
            if self.mode_ac:
                self.demodulate_ac(data, n, sample_ts, sys_ts, mean_level, mean_power)
            self.filter.expire(self.ifile_now)  # backgroundTasks() after every buffer, mstime() = Modes.ifile_now
            counter += n
            if n < BUF:  # a short (or empty) read is the end of the file
                break
        return self.messages, self.stats

    # ---- a live receiver's feed: rtlsdrCallback's sample clock, which runs on over blocks dropped for want of a free
    # buffer (sdr_rtlsdr.c:279-301), the MAGBUF_DISCONTINUOUS buffer behind such a gap with zeros for its look-behind
    # (fifo.c:179-185), --ifile's system clock (sdr_ifile.c:190).  segments: raw IQ, all but the last whole buffers;
    # drops[i]: samples lost in front of segment i.  The last segment ends like a file.
    def live_feed(self, segments, drops):
        assert not self.dc_filter
        bps = 2 if self.fmt == "uc8" else 4
        tail = np.zeros(OVERLAP, dtype=np.uint16)
        counter = 0
        for si, (seg, drop) in enumerate(zip(segments, drops)):
            raw = bytes(seg)
            nsamples = len(raw) // bps
            last = si == len(segments) - 1
            assert last or nsamples % BUF == 0
            counter += drop
            discontinuous = drop > 0
            off = 0
            while True:
                n = min(BUF, nsamples - off)
                if n == 0 and not last:
                    break
                mag, lvl, pwr, float_sums = convert(self.fmt, raw[off * bps: (off + n) * bps], False)
                data = np.concatenate([np.zeros(OVERLAP, dtype=np.uint16) if discontinuous else tail, mag])
                discontinuous = False
                sample_ts = int(counter * 12e6 / 2.4e6)
                sys_ts = sample_ts // 12000 + self.startup_time
                mean_level, mean_power = buffer_means(float_sums, lvl, pwr)
                tail = data[len(data) - OVERLAP:]
                self.demodulate(data, n, sample_ts, sys_ts, mean_power)
                if self.mode_ac:
                    self.demodulate_ac(data, n, sample_ts, sys_ts, mean_level, mean_power)
                self.filter.expire(self.ifile_now)
                counter += n
                off += n
                if n < BUF:
                    break
        return self.messages, self.stats
