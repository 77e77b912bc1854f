"""An input corpus that owes nothing to csrc/msd_siggen.c: pure-numpy 1090 MHz baseband captures.

Everything is built from the public signal definition, not from the generator the product ships with:
  * Mode S frames (ICAO Annex 10 vol IV 3.1.2): 8 us preamble with pulses at 0, 1, 3.5 and 4.5 us, then 56 or 112
    bits of PPM at 1 us per bit (pulse in the first half = 1, in the second half = 0), parity from a bitwise long
    division by the generator polynomial 0xFFF409 written here;
  * Mode A/C replies (Annex 10 vol IV 3.1.1.6): 0.45 us pulses on a 1.45 us grid, F1 ... F2 20.3 us apart, the twelve
    code pulses C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4 in between;
  * a 12 MHz envelope grid (five ticks per 2.4 MSPS sample), per-frame amplitude, carrier phase and carrier frequency
    offset, box-averaged to the sample grid, Gaussian I/Q noise, optional DC offset, clipping and a delayed echo;
  * quantised to UC8 (rtl-sdr: 127.5 + 127.5 x), SC16 (x 32767) or SC16Q11 (x 2047).
Returns the capture and the list of frames put into it (start sample, payload, amplitude), so that a test can ask two
independent questions: does the GPU path equal the oracle, and does it find the frames that were transmitted."""
import numpy as np

TICKS = 5  # 12 MHz ticks per 2.4 MSPS sample
GENERATOR = 0xFFF409


def crc24(bits):
    """Remainder of bits(x) * x^24 divided by the generator: plain long division over GF(2) of the data bits followed
    by 24 zero bits."""
    work = np.concatenate([np.asarray(bits, dtype=np.uint8), np.zeros(24, dtype=np.uint8)])
    gen = to_bits((1 << 24) | GENERATOR, 25)
    for i in range(len(bits)):
        if work[i]:
            work[i:i + 25] ^= gen
    return int("".join(str(int(b)) for b in work[-24:]), 2)


def to_bits(value, n):
    return np.array([(value >> (n - 1 - i)) & 1 for i in range(n)], dtype=np.uint8)


def mode_s_frame(rng, df, address, flip=None):
    """Bits of a DF `df` frame from `address`: DF17 / DF11 carry the address in the clear and plain parity, the others
    (address/parity) have the address xored onto the parity."""
    long_frame = df >= 16
    nbits = 112 if long_frame else 56
    body = np.concatenate([to_bits(df, 5), rng.integers(0, 2, size=nbits - 5 - 24, dtype=np.uint8)])
    if df in (11, 17, 18):
        body[8:32] = to_bits(address, 24)
        if df == 11:
            pass  # II = 0: plain parity
    parity = crc24(body)
    if df not in (11, 17, 18):
        parity ^= address
    bits = np.concatenate([body, to_bits(parity, 24)])
    if flip is not None:
        bits[flip] ^= 1
    return bits


def mode_s_envelope(bits):
    """Unit envelope on the 12 MHz grid: 6 ticks per half microsecond."""
    env = np.zeros((8 + len(bits)) * 12, dtype=np.float32)
    for start_us2 in (0, 2, 7, 9):  # preamble pulses at 0, 1.0, 3.5, 4.5 us, in half microseconds
        env[start_us2 * 6:(start_us2 + 1) * 6] = 1.0
    for i, b in enumerate(bits):
        t0 = (8 + i) * 12 + (0 if b else 6)
        env[t0:t0 + 6] = 1.0
    return env


def mode_ac_envelope(code12):
    """F1, the twelve code pulses and X (never set), F2: 0.45 us pulses, 1.45 us apart -- on the 12 MHz grid 5.4 ticks
    wide and 17.4 ticks apart, rendered with fractional edges."""
    n = int(np.ceil(15 * 17.4)) + 8
    env = np.zeros(n, dtype=np.float32)
    present = [1] + [(code12 >> (11 - i)) & 1 for i in range(6)] + [0] + [(code12 >> (5 - i)) & 1 for i in range(6)] + [1]
    for slot, on in enumerate(present):
        if not on:
            continue
        a, b = slot * 17.4, slot * 17.4 + 5.4
        for t in range(int(a), int(np.ceil(b))):
            env[t] += min(b, t + 1) - max(a, t)
    return env


def capture(seed, nsamples, fmt="uc8", frames_per_sec=1500.0, ac_per_sec=0.0, noise=0.02, n_aircraft=60, amp=(0.08, 0.9),
            freq_offset_hz=0.0, dc=(0.0, 0.0), clip_gain=1.0, echo=None, flip_fraction=0.03, random_bytes=False):
    """-> (iq uint8 array, frames): frames is a list of dicts (sample, bits, amp, df, addr, kind)."""
    rng = np.random.default_rng(seed)
    bps = 2 if fmt == "uc8" else 4
    if random_bytes:
        return rng.integers(0, 256, size=nsamples * bps, dtype=np.uint8), []
    nt = nsamples * TICKS
    sig = np.zeros(nt, dtype=np.complex64)
    addrs = rng.integers(1, 1 << 24, size=n_aircraft)
    frames = []
    t = 0.0
    mean_gap = 12e6 / max(frames_per_sec, 1e-9)
    while frames_per_sec > 0:
        t += rng.exponential(mean_gap)
        start = int(t)
        if start + 1600 >= nt:
            break
        df = int(rng.choice([17, 17, 17, 17, 11, 11, 4, 5, 20, 21, 0, 16]))
        addr = int(rng.choice(addrs))
        nbits = 112 if df >= 16 else 56
        flip = int(rng.integers(5, nbits)) if rng.random() < flip_fraction else None
        bits = mode_s_frame(rng, df, addr, flip)
        a = float(np.exp(rng.uniform(np.log(amp[0]), np.log(amp[1]))))
        env = mode_s_envelope(bits)
        ph = rng.uniform(0, 2 * np.pi) + 2 * np.pi * freq_offset_hz * (np.arange(env.size) / 12e6) * rng.choice([-1.0, 1.0])
        sig[start:start + env.size] += (a * env * np.exp(1j * ph)).astype(np.complex64)
        frames.append(dict(tick=start, bits=bits, amp=a, df=df, addr=addr, kind="S", flipped=flip is not None))
    t = 0.0
    mean_gap = 12e6 / max(ac_per_sec, 1e-9)
    while ac_per_sec > 0:
        t += rng.exponential(mean_gap)
        start = int(t)
        if start + 400 >= nt:
            break
        code = int(rng.integers(0, 4096))
        a = float(np.exp(rng.uniform(np.log(max(amp[0], 0.2)), np.log(amp[1]))))
        env = mode_ac_envelope(code)
        sig[start:start + env.size] += (a * env * np.exp(1j * rng.uniform(0, 2 * np.pi))).astype(np.complex64)
        frames.append(dict(tick=start, code=code, amp=a, kind="AC"))
    if echo is not None:  # (delay in ticks, relative amplitude, phase)
        d, g, p = echo
        sig[d:] += (g * np.exp(1j * p)) * sig[:-d].copy()
    x = sig.reshape(nsamples, TICKS).mean(axis=1)  # the receiver's anti-alias + decimation, as a box filter
    x = x + (rng.normal(0, noise, nsamples) + 1j * rng.normal(0, noise, nsamples)).astype(np.complex64)
    x = x * clip_gain + (dc[0] + 1j * dc[1])
    i, q = np.real(x), np.imag(x)
    if fmt == "uc8":
        out = np.empty(2 * nsamples, dtype=np.uint8)
        out[0::2] = np.clip(np.rint(127.5 + 127.5 * i), 0, 255).astype(np.uint8)
        out[1::2] = np.clip(np.rint(127.5 + 127.5 * q), 0, 255).astype(np.uint8)
        return out, frames
    full = 32767 if fmt == "sc16" else 2047
    out = np.empty(2 * nsamples, dtype=np.int16)
    out[0::2] = np.clip(np.rint(full * i), -full - 1, full).astype(np.int16)
    out[1::2] = np.clip(np.rint(full * q), -full - 1, full).astype(np.int16)
    return out.view(np.uint8), frames


def frame_bytes(bits):
    return np.packbits(bits).tobytes()


def isolated_strong_frames(frames, min_amp=0.25, guard_ticks=2200):
    """Mode S frames that a receiver has no excuse to miss: strong, unflipped, nothing else on the air around them."""
    ticks = np.array([f["tick"] for f in frames])
    out = []
    for k, f in enumerate(frames):
        if f["kind"] != "S" or f["flipped"] or f["amp"] < min_amp:
            continue
        near = np.abs(ticks - f["tick"]) < guard_ticks
        if near.sum() == 1:
            out.append(f)
    return out
