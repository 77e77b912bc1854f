"""ctypes binding of the seeded synthetic-capture generator (csrc/msd_siggen.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "msd_siggen.c")
_LIB = os.path.join(_HERE, "csrc", "libmsd_siggen.so")

UC8, SC16, SC16Q11 = 0, 1, 2
BLOCK = 4096


class Cfg(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("format", C.c_uint32),
        ("slot_samples", C.c_uint32),
        ("ac_slot_samples", C.c_uint32),
        ("noise_q16", C.c_uint32),
        ("n_aircraft", C.c_uint32),
        ("flip_permille", C.c_uint32),
        ("overlap_permille", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


def build(force=False):
    hdr = os.path.join(_HERE, "csrc", "msd_siggen.h")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC),
                                                                         os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-o", _LIB,
                               _SRC, "-lpthread"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.msd_siggen_generate.restype = C.c_int
        L.msd_siggen_generate.argtypes = [C.POINTER(Cfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint]
        L.msd_siggen_aircraft.restype = C.c_uint32
        L.msd_siggen_aircraft.argtypes = [C.POINTER(Cfg), C.c_uint32]
        _lib = L
    return _lib


def make_cfg(seed, fmt=UC8, msgs_per_sec=2000, ac_per_sec=0, noise_fs=0.02, n_aircraft=50,
             flip_permille=20, overlap_permille=10):
    """Content model of SURVEY.md 8(d): rates are converted to slot lengths at 2.4 MSPS."""
    return Cfg(seed=seed, format=fmt,
               slot_samples=int(round(2400000 / msgs_per_sec)) if msgs_per_sec else 0,
               ac_slot_samples=int(round(2400000 / ac_per_sec)) if ac_per_sec else 0,
               noise_q16=int(round(noise_fs * 65536)), n_aircraft=n_aircraft,
               flip_permille=flip_permille, overlap_permille=overlap_permille, reserved=0)


def bytes_per_sample(fmt):
    return 2 if fmt == UC8 else 4


def generate(cfg, nsamples, first_sample=0, out=None, nthreads=None):
    """Return a uint8 array holding nsamples IQ samples starting at first_sample (multiple of 4096)."""
    bps = bytes_per_sample(cfg.format)
    if out is None:
        out = np.empty(nsamples * bps, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= nsamples * bps and out.flags.c_contiguous
    if nthreads is None:
        nthreads = min(32, os.cpu_count() or 1)
    rc = lib().msd_siggen_generate(C.byref(cfg), first_sample, nsamples, out.ctypes.data, nthreads)
    if rc != 0:
        raise ValueError(f"msd_siggen_generate failed: {rc}")
    return out
