"""The C-ABI library loads on a machine without a GPU, exports every symbol include/modes_hip.h
declares, and refuses to work without a device (no CPU fallback)."""
import ctypes as C
import errno
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "modes_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msd_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("msd_create", "msd_destroy", "msd_submit_device", "msd_submit_host", "msd_launch_device",
                 "msd_collect", "msd_convert", "msd_demodulate_magbuf", "msd_get_stats", "msd_reset"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.lib()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in modes_hip.h but not exported"
    assert sorted(pkg.capi.EXPORTS) == declared_functions()


def declared_host_functions():
    text = open(os.path.join(ROOT, "include", "modes_hip_readsb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msd_[A-Za-z_0-9]+)\s*\(", text)))


def test_host_boundary_library_exports(pkg):
    """libmsd_host.so exports everything include/modes_hip_readsb.h declares: the converter factory
    (convert.h:40-45), the mag_buf FIFO (fifo.h:80-120) and the ifile handler (sdr.c:41-50)."""
    host = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    names = declared_host_functions()
    for must in ("msd_init_converter", "msd_cleanup_converter", "msd_fifo_create", "msd_fifo_destroy", "msd_fifo_drain",
                 "msd_fifo_halt", "msd_fifo_acquire", "msd_fifo_enqueue", "msd_fifo_dequeue", "msd_fifo_release",
                 "msd_ifileInitConfig", "msd_ifileHandleOption", "msd_ifileOpen", "msd_ifileRun", "msd_ifileClose",
                 "msd_ifileSetOptionKeys", "msd_ifileSetHooks"):
        assert must in names, must
    for name in names:
        assert hasattr(host, name), f"{name} is declared in modes_hip_readsb.h but not exported"


def test_converter_factory_without_a_gpu_returns_null_like_the_reference(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    host = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    host.msd_init_converter.restype = C.c_void_p
    host.msd_init_converter.argtypes = [C.c_int, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
    state = C.c_void_p(1)
    assert host.msd_init_converter(0, 2400000.0, 0, C.byref(state)) is None and state.value is None


def test_message_struct_layout_matches_numpy_mirror(pkg, oracle):
    assert pkg.MESSAGE_DTYPE == oracle.MESSAGE_DTYPE
    assert pkg.MESSAGE_DTYPE.itemsize == 56
    assert pkg.MESSAGE_DTYPE.fields["msg"][1] == 40 and pkg.MESSAGE_DTYPE.fields["iid"][1] == 54


def test_no_cpu_fallback_without_a_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.MsdError) as e:
        pkg.Demodulator()
    assert str(-errno.ENODEV) in str(e.value) or "No such device" in str(e.value)


def test_bad_configuration_is_rejected(pkg):
    lib = pkg.capi.lib()
    h = C.c_void_p()
    for kw in (dict(format=9), dict(nfix_crc=3), dict(preamble_threshold=0)):
        cfg = pkg.capi.Config(device=0, format=0, preamble_threshold=58, nfix_crc=1, mode_ac=0, reserved0=0,
                              max_batch_samples=131072, stream=None)
        for k, v in kw.items():
            setattr(cfg, k, v)
        assert lib.msd_create(C.byref(cfg), C.byref(h)) == -errno.EINVAL
    assert lib.msd_create(None, C.byref(h)) == -errno.EINVAL


def test_sc16q11_table_bits_outside_the_range_are_refused(pkg):
    """The reference's SC16Q11_TABLE_BITS is a compile-time constant; the setter that stands for it says -EINVAL for a value
    no build could have (it used to fall back to the float path silently, while msd_create refused the same value)."""
    host = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    host.msd_converter_set_sc16q11_table_bits.restype = C.c_int
    assert host.msd_converter_set_sc16q11_table_bits(12) == -errno.EINVAL
    assert host.msd_converter_set_sc16q11_table_bits(-1) == -errno.EINVAL
    assert host.msd_converter_set_sc16q11_table_bits(8) == 0 and host.msd_converter_set_sc16q11_table_bits(0) == 0


@pytest.mark.parametrize("nfix_crc", [0, 1, 2])
def test_host_built_tables_pass_their_selftest(pkg, nfix_crc):
    """msd_tables_build + msd_tables_selftest on the host (msd_create runs the same check before it uploads anything): the
    folded UC8 table and the scan kernel's 256-pitch swizzled copy against the reference's 65536-entry table (convert.c:35-61),
    every single-bit syndrome in its bucket of four (crc.c:367-412), the slicer tables against the closed form of
    demod_2400.c:98-177, the per-group syndromes against modesChecksum."""
    lib = pkg.capi.lib()
    buf = C.create_string_buffer(1 << 20)            # sizeof(msd_tables) is about 0.4 MB
    lib.msd_tables_build.restype = None
    lib.msd_tables_build.argtypes = [C.c_void_p, C.c_int]
    lib.msd_tables_selftest.restype = C.c_int
    lib.msd_tables_selftest.argtypes = [C.c_void_p]
    lib.msd_tables_build(buf, nfix_crc)
    assert lib.msd_tables_selftest(buf) == 0
    tail = bytes(buf)[(1 << 20) - 4096:]
    assert tail == b"\0" * 4096                       # the struct fits the buffer with room to spare
