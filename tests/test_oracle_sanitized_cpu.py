"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's CI would run its native
code under sanitizers).  oracle/corsair_oracle.c decides every parity test, so its own index arithmetic is
checked here: the oracle's CPU test cases run again in a child interpreter on the instrumented build
(`ORACLE_SANITIZE=1`: -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined), with gcc's libasan preloaded.
CPU only -- sanitizer builds never go to the GPU box (GPU ASan is not available on the pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every test that drives the C oracle (sparse conv / kernel maps / forward, k-NN, Chamfer, rigid fit, RANSAC, part cut,
# voxel index) + the NumPy layers on top of it
CASES = "sparse_conv or real_cloud or knn_and_chamfer or rigid_fit or ransac_oracle or symmetric_cut or sym_pose_host or quantize"


def test_c_oracle_is_clean_under_asan_and_ubsan():
    from oracle import native

    asan = native.sanitizer_runtime()
    assert os.path.exists(asan), "gcc's libasan.so is needed for the sanitizer build"
    env = dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=asan,
               # python itself is not instrumented: its arena allocator "leaks" by design; the oracle's own mallocs are
               # paired in-function and still checked for overflow / use-after-free
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", ORACLE_THREADS="8")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_cpu.py"),
                        os.path.join(ROOT, "tests", "test_pins_cpu.py"), "-x", "-q", "-k", CASES + " or kmeans or pins or BatchNorm",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (p.stdout[-3000:] + "\n" + p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, tail
    assert p.returncode == 0, tail
    assert " passed" in p.stdout, tail
    # the instrumented library really was the one loaded
    assert os.path.exists(os.path.join(ROOT, "oracle", "_build", "libcorsair_oracle_san.so"))
