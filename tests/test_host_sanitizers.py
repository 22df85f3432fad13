"""CPU-only: the HOST side of librender_mi355x.so under AddressSanitizer + UndefinedBehaviorSanitizer
and ThreadSanitizer.  Sanitizers run on the CPU build only (GPU ASan is not available on this pool): csrc/host_helpers.cpp
(grid builder, MT19937 windows, gen_rays, scene generators, contexts), csrc/pt_leaf.h (pairwise-sum plans) and the oracle's C
restatement are compiled from source into throw-away drivers (tests/sanitize/*.cpp) that also CHECK what they compute."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ascendpathtracing_amd", "csrc")
FP = ["-ffp-contract=off", "-fno-fast-math"]          # the product's own arithmetic flags (csrc/Makefile)


def _build(tmp_path, name, sanitize, sources, extra=()):
    exe = str(tmp_path / name)
    objs = []
    for src in sources:
        obj = str(tmp_path / (os.path.basename(src) + "." + name + ".o"))
        cc = ["gcc", "-std=c11", "-fopenmp"] if src.endswith(".c") else ["g++", "-std=c++17"]
        subprocess.run(cc + ["-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitize}", "-fno-sanitize-recover=all", *FP, *extra,
                             "-c", src, "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++", f"-fsanitize={sanitize}", "-fopenmp", "-o", exe, *objs, "-lm", "-lpthread"], check=True)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_helpers_and_oracle_under_asan_and_ubsan(tmp_path):
    exe = _build(tmp_path, "host_driver", "address,undefined",
                 [os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp"), os.path.join(CSRC, "host_helpers.cpp"),
                  os.path.join(ROOT, "oracle", "pt_oracle.c")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert r.stdout.startswith("ok ") and int(r.stdout.split()[1]) > 100000      # the driver's own checks ran
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_contexts_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "context_threads", "thread",
                 [os.path.join(ROOT, "tests", "sanitize", "context_threads.cpp"), os.path.join(CSRC, "host_helpers.cpp")])
    probe = subprocess.run([exe], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    if "unexpected memory mapping" in probe.stderr or "ThreadSanitizer: failed to" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container: " + probe.stderr.strip().splitlines()[0])
    assert probe.returncode == 0, (probe.stdout[-300:], probe.stderr[-3000:])
    assert probe.stdout.strip() == "ok" and "WARNING: ThreadSanitizer" not in probe.stderr, probe.stderr[-3000:]
