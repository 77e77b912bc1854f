"""The threaded and the plain host C under AddressSanitizer + UBSan and under ThreadSanitizer (the
reference builds with neither, Makefile:13, and carries fifo.c:141,192-197,219).  scripts/sanitize.sh builds the
instrumented copies into build/san_{asan,tsan}/; zero reports is the bar:

  CPU  tests/c/fifo_stress.c   producer twelve buffers ahead, consumer checking order / overlap / content, a halt in
                               mid-stream from a third thread (both sanitizers)
       tests/c/host_units.c    wire formats, tables + self-check + both repair tables, field decoder on every DF with
                               extreme payloads, pacer, error paths of the ifile handler and the converter factory
       the host-side python tests (FIFO against the reference's own fifo.c, throttle, wire formats, fields, ABI,
       boundary) against the ASan build, the runtime preloaded into the interpreter
  GPU  the replay tool, both sanitizers, both paths: reader thread + consumer thread + the context's helper thread
       (fused), and the mag_buf path with the host resolver's thread pool
(ThreadSanitizer cannot be preloaded into this python -- it never gets past interpreter start-up --, so its share is
the C drivers and the replay tool; that build is the ROCm clang's throughout, the C++ launcher msd_capi.cpp included.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_TESTS = ["tests/test_host_fifo.py", "tests/test_throttle.py", "tests/test_wire_formats.py", "tests/test_fields.py", "tests/test_abi.py",
              "tests/test_boundary.py::test_boundary_compiles_against_reference_style_declarations_and_runs"]


def gcc_file(name):
    return subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()


@pytest.fixture(scope="module", params=["asan", "tsan"])
def san(request, pkg):
    mode = request.param
    out = os.path.join(ROOT, "build", "san_" + mode)
    subprocess.check_call(["bash", os.path.join(ROOT, "scripts", "sanitize.sh"), mode, out], stdout=subprocess.DEVNULL)
    sym = "__asan_init" if mode == "asan" else "__tsan_init"
    for lib in ("libmsd_host.so", "libmodes_hip.so"):   # the instrumentation is really in there
        assert sym in subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(out, lib)], text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=66",
               TSAN_OPTIONS="exitcode=66:halt_on_error=0")
    return mode, out, env


def no_aslr(mode):
    """(the ThreadSanitizer build is clang's: its runtime knows the GPU boxes' address-space layout; nothing to switch off)"""
    return []


def run_clean(cmd, env, timeout=600, **kw):
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, **kw)
    text = res.stdout + res.stderr
    assert res.returncode == 0 and "Sanitizer" not in text and "runtime error" not in text, text[-4000:]
    return text


def test_fifo_stress_under_the_sanitizers(san):
    mode, out, env = san
    assert "fifo stress ok" in run_clean(no_aslr(mode) + [os.path.join(out, "fifo_stress")], env)


def test_host_units_under_the_sanitizers(san):
    mode, out, env = san
    assert "host units ok" in run_clean(no_aslr(mode) + [os.path.join(out, "host_units")], env)


def test_host_side_python_tests_against_the_asan_build(san):
    mode, out, env = san
    if mode != "asan":
        pytest.skip("ThreadSanitizer cannot be preloaded into the interpreter; its share is the C drivers and the replay tool")
    env = dict(env, MSD_LIBMODES_HIP=os.path.join(out, "libmodes_hip.so"), LD_PRELOAD=gcc_file("libasan.so") + ":" + gcc_file("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:exitcode=66")   # (the interpreter's own allocations are not ours to account for)
    text = run_clean([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + HOST_TESTS, env, cwd=ROOT)
    assert " passed" in text and "failed" not in text, text[-2000:]


@pytest.mark.gpu
def test_replay_tool_under_the_sanitizers_on_the_gpu(san, torch_cuda, tmp_path):
    """Both run modes of the handler with their threads (host/msd_sdr_ifile.c: reader + consumer; msd_capi.cpp's helper
    thread; msd_resolve.c's pool behind the mag_buf path) on a capture with traffic; the instrumented tool prints what
    the ordinary one prints."""
    mode, out, env = san
    n = 40 * pkg_chunk() + 777
    import __graft_entry__ as graft
    P = graft.load_package()
    iq = P.siggen.generate(P.siggen.make_cfg(seed=4242), n)
    cap = tmp_path / "cap.uc8"
    iq.tofile(cap)
    plain = os.path.join(os.path.dirname(P.capi.LIB_PATH), "msd_replay")
    # HIP's own runtime threads are not instrumented: races the tool would report inside them are not the host C's
    supp = tmp_path / "tsan.supp"
    supp.write_text("called_from_lib:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\nrace:libamdhip64.so\nrace:libhsa-runtime64.so\n")
    env = dict(env, TSAN_OPTIONS=env["TSAN_OPTIONS"] + ":suppressions=" + str(supp), ASAN_OPTIONS="detect_leaks=0:exitcode=66:protect_shadow_gap=0")
    for path in ("fused", "magbuf"):
        args = ["--ifile", str(cap), "--iformat", "uc8", "--fix", "--path", path, "--stats"]
        want = subprocess.run([plain] + args, capture_output=True, text=True, timeout=600)
        assert want.returncode == 0, want.stderr[-2000:]
        got = subprocess.run(no_aslr(mode) + [os.path.join(out, "msd_replay")] + args, capture_output=True, text=True, timeout=900, env=env)
        text = got.stdout + got.stderr
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):   # the full reports, for whoever has to read them
            with open(os.path.join(ROOT, "gpurun_out", "sanitizer_%s_%s.txt" % (mode, path)), "w") as f:
                f.write(got.stderr)
        assert got.returncode == 0 and "Sanitizer" not in text and "runtime error" not in text, (mode, path, text[-4000:])
        assert got.stdout == want.stdout and got.stdout.count("\n") > 100, (mode, path)


def pkg_chunk():
    import __graft_entry__ as graft
    return graft.load_package().CHUNK
