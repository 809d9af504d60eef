"""Sanitizer runs of the product's threaded C++ host code (CPU only — GPU ASan is not available on the
pool): `make -C sharkmer_amd/csrc san` builds csrc/shk_front.cpp + shk_inflate.cpp — the FASTQ front-end with its
producer / inflate / parse-pool / copy-pool threads, the packer, the writers — with -fsanitize=address,undefined and
again with -fsanitize=thread, and the CPU tests of that code run against each build in a child interpreter
(SHK_FRONT_LIB points the ctypes mirror at the sanitized library; the sanitizer runtime is preloaded).  Any report
fails the run: ASan/UBSan abort (halt_on_error), TSan exits with 66.  ThreadSanitizer cannot be preloaded into a
Python process, so its run goes through a native caller of the same C ABI (tests/native/front_driver.cpp)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sharkmer_amd", "csrc")
TESTS = ["tests/test_host_cpu.py", "tests/test_packed_cpu.py", "tests/test_frontend_semantics.py"]


def _runtime(name):
    out = subprocess.run(["g++", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.fixture(scope="module")
def san_libs():
    subprocess.check_call(["make", "-C", CSRC, "san"], stdout=subprocess.DEVNULL)
    import __graft_entry__ as g
    g.build()   # (the children find everything built: a build under a preloaded sanitizer runtime is no test of ours)


def test_host_tests_clean_under_asan_ubsan(san_libs, tmp_path):
    rt = _runtime("libasan.so")
    if rt is None:
        pytest.skip("no asan runtime in this toolchain")
    env = dict(os.environ)
    env["SHK_FRONT_LIB"] = os.path.join(CSRC, "libshk_front_asan.so")
    env["LD_PRELOAD"] = rt
    env["SHK_FUZZ_SEEDS"] = "3"
    log = str(tmp_path / "san")
    # (leak checking off: the interpreter itself never frees everything; every reader handle here is closed)
    env["ASAN_OPTIONS"] = f"detect_leaks=0:halt_on_error=1:abort_on_error=1:log_path={log}"
    env["UBSAN_OPTIONS"] = f"halt_on_error=1:print_stacktrace=1:log_path={log}"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *TESTS], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    reports = [f for f in os.listdir(tmp_path) if f.startswith("san")]
    detail = "".join(open(os.path.join(tmp_path, f)).read()[:4000] for f in reports)
    assert r.returncode == 0 and not reports, (r.returncode, r.stdout[-3000:], r.stderr[-2000:], detail)
    assert " passed" in r.stdout


def _fastq(rng, n, bad=None):
    recs = []
    for i in range(n):
        L = int(rng.integers(0, 200))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        recs.append(f"@r{i} some text\n{seq}\n+\n{'I' * L}\n")
    if bad is not None:
        recs[bad] = recs[bad].replace("+", "-", 1)
    return "".join(recs).encode()


def test_threaded_reader_clean_under_tsan(san_libs, tmp_path):
    """ThreadSanitizer cannot be preloaded into the interpreter, so a native caller (tests/native/front_driver.cpp) drives
    the same entry points: several files at once — plain, gzip, gzip by magic, one truncated — through small windows and
    many threads (producers running ahead, the inflate thread, both pools), then the packer; and again with every gzip
    member forced through the many-thread decoder (speculating workers, the stitcher, the files' turns).  No report, and the same
    digest as the unsanitized build."""
    import zlib
    import numpy as np
    if _runtime("libtsan.so") is None:
        pytest.skip("no tsan runtime in this toolchain")
    rng = np.random.default_rng(3)

    def gz(b):
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        return co.compress(b) + co.flush()

    files = []
    for i, (n, kind) in enumerate([(30000, "plain"), (20000, "gz"), (1, "plain"), (15000, "magic"), (9000, "plain")]):
        data = _fastq(rng, n)
        p = tmp_path / (f"f{i}.fastq" + (".gz" if kind == "gz" else ""))
        p.write_bytes(data if kind == "plain" else gz(data))
        files.append(str(p))
    flawed = tmp_path / "flawed.fastq.gz"
    flawed.write_bytes(gz(_fastq(rng, 5000, bad=4000)))
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(gz(_fastq(rng, 5000))[:30000])
    env0 = dict(os.environ)
    env0.update(SHK_FASTQ_WINDOW_KB="256", SHK_FASTQ_THREADS="6", SHK_FASTQ_COPY_THREADS="4")
    log = str(tmp_path / "tsan")
    env0["TSAN_OPTIONS"] = f"halt_on_error=0:exitcode=66:log_path={log}"
    pgz = dict(SHK_PGZ_MIN_KB="0", SHK_PGZ_CHUNK_KB="16", SHK_PGZ_THREADS="5")   # every member through the many-thread decoder, files in turn
    for args, extra in ((["0", "0", "20000", "3000000"] + files, {}),
                        (["61234", "7", "5000", "1000000"] + files, {}),
                        (["0", "1000", "100000", "30000000"] + files[:2] + [str(flawed)] + files[2:], {}),
                        (["0", "0", "100000", "30000000"] + files[:3] + [str(cut)], {}),
                        (["0", "0", "20000", "3000000"] + files, pgz),
                        (["9000", "0", "5000", "1000000"] + files, pgz),
                        (["0", "0", "100000", "30000000"] + files[:3] + [str(cut)] + [str(flawed)], pgz)):
        env = dict(env0)
        env.update(extra)
        want = subprocess.run([os.path.join(CSRC, "front_driver_plain"), *args], env=env, capture_output=True, text=True, timeout=300)
        got = subprocess.run([os.path.join(CSRC, "front_driver_tsan"), *args], env=env, capture_output=True, text=True, timeout=600)
        reports = [f for f in os.listdir(tmp_path) if f.startswith("tsan.")]
        detail = "".join(open(os.path.join(tmp_path, f)).read()[:6000] for f in reports)
        assert got.returncode == 0 and not reports, (got.returncode, got.stdout, got.stderr[-2000:], detail)
        assert got.stdout == want.stdout and got.stdout.startswith(("ok reads", "error")), (got.stdout, want.stdout)
