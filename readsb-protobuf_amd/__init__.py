"""MI355X-native Mode S / Mode A/C receive path (readsb 2.4 MSPS hot path) -- Python host mirror.

The product is csrc/libmodes_hip.so (hand-written gfx950 HIP kernels behind the C-ABI of
include/modes_hip.h).  This package only binds it; torch is used by bench.py/tests for device
memory and streams, never for compute.
"""
from . import capi, sharding, siggen  # noqa: F401
from .capi import (CHUNK, FMT_MAG16, FMT_SC16, FMT_SC16Q11, FMT_UC8, MESSAGE_DTYPE, OVERLAP, Demodulator,  # noqa: F401
                   MsdError, replay_device)
