"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has
none; `make -C oracle asan`).  The golden-vector and primitive suites are re-run in a child interpreter with the
sanitizer runtimes preloaded and ORACLE_LIB pointing at the instrumented build; any report fails the run.
CPU only: the GPU pool offers no sanitizers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suites_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",     # CPython itself leaks by design
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_golden.py"), os.path.join(ROOT, "tests", "test_oracle_primitives.py"),
                        os.path.join(ROOT, "tests", "test_vibrato.py"),
                        "-m", "not gpu"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "passed" in r.stdout and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]


def test_host_file_writer_under_asan_ubsan(tmp_path):
    """The one host-only translation unit of the product (csrc/fileio.cpp: the native writer of the sweep's feature
    files, plain threads) built by g++ with AddressSanitizer + UndefinedBehaviorSanitizer around a harness
    (tests/hostsan/fileio_harness.cpp): many files from several threads, empty files, more threads than files, bad
    arguments, a path that cannot be created."""
    if not _runtime("libasan.so") or not _runtime("libubsan.so"):
        pytest.skip("gcc sanitizer runtimes not installed")
    exe = tmp_path / "fileio_harness"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "hostsan", "fileio_harness.cpp"),
                           os.path.join(ROOT, "hts-train-world_amd", "csrc", "fileio.cpp"), "-o", str(exe)])
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    r = subprocess.run([str(exe), str(out_dir)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="abort_on_error=1:halt_on_error=1",
                                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert r.returncode == 0 and "fileio ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_host_layer_of_the_c_abi_under_asan_ubsan(tmp_path):
    """the host marshalling under sanitizers.  Every translation unit of the library is compiled
    host-only (hipcc --cuda-host-only: no device code) with AddressSanitizer + UBSan and linked against
    tests/hostsan/hip_stub.cpp -- "device" memory is host heap, kernels do not run -- and tests/hostsan/capi_harness.cpp
    calls WORLD's C ABI like the reference CLIs: `double**` rows allocated one by one at their exact size, ten
    utterance shapes (8 ... 48 kHz, 1 / 5 / 10 ms, non-default fft sizes, Harvest, 7 frames), the codec entry points,
    refused arguments through the error handler, a box without a device, and device-allocation failures injected at a
    dozen points of the run (every one must reach the handler; nothing may leak or touch freed memory)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rt = [p for p in ("/opt/rocm/lib/llvm/lib/clang",) if os.path.isdir(p)]
    if not os.path.exists(hipcc) or not rt:
        pytest.skip("hipcc / clang sanitizer runtimes not installed")
    out = tmp_path / "hs"
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "hostsan", "build_capi_harness.sh"), str(out)],
                       capture_output=True, text=True, timeout=900)
    exe = out / "capi_harness"
    assert r.returncode == 0 and exe.exists(), (r.stdout[-2000:], r.stderr[-2000:],
                                                  [open(p).read()[-600:] for p in map(str, out.glob("*.log")) if os.path.getsize(p)])
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:halt_on_error=1:detect_leaks=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")

    def run(**extra):
        p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(env, **extra))
        text = p.stdout + p.stderr
        assert p.returncode == 0, text[-4000:]
        assert "AddressSanitizer" not in text and "LeakSanitizer" not in text and "runtime error" not in text, text[-4000:]
        return text

    text = run()
    assert "capi ok" in text and "device allocations" in text
    assert "capi no-device ok" in run(HIP_STUB_DEVICES="0")
    for n in (0, 1, 3, 8, 21, 34, 55, 89, 120):
        assert "handler:" in run(HIP_STUB_FAIL_MALLOC_AFTER=str(n))
