"""Sanitizer builds on the CPU box ("race detection / sanitizers": compile-time -fsanitize=address on the
host side; GPU AddressSanitizer is not available on this pool).

  * oracle/textcnn_ref.c (the C restatement the GPU parity tests trust) built with gcc -fsanitize=address,undefined, and
    tests/test_c_oracle.py run against that build;
  * the HOST side of the product library -- descriptor validation, channel-tile plans, workspace layouts and every
    *_ws_bytes / *_ws_floats / *_partial_elems / *_packed_floats size function of csrc/*.hip -- built with
    hipcc --offload-host-only -fsanitize=address,undefined and fuzzed with hostile descriptors and NULL arguments
    (tests/asan_fuzz_host.py): every call must end in a size, 0, RBR_ERR_BAD_ARG or RBR_ERR_UNSUPPORTED, never in a
    sanitizer report.

`make -C oracle asan` builds both (seconds).  The first run of this fuzz found prod_applicable() reading d->kz[8] of a descriptor
with n_widths > 8 before validation, and signed overflows in rbr_textcnn_taps_owner_rows / rbr_review_attn_bwd_ws_floats."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE = os.path.join(ROOT, "oracle")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:alloc_dealloc_mismatch=0:detect_odr_violation=0",
           "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


def _make_asan():
    if shutil.which("make") is None:
        pytest.skip("no make on this box")
    r = subprocess.run(["make", "-s", "-j4", "-C", ORACLE, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_c_oracle_under_address_and_ub_sanitizer():
    _make_asan()
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc has no shared ASan runtime here")
    env = dict(os.environ, LD_PRELOAD=rt, RBR_C_REF_LIB=os.path.join(ORACLE, "_build", "libtextcnn_ref_asan.so"), **SAN_ENV)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_c_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "passed" in r.stdout, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]


def test_host_side_of_the_c_abi_under_address_and_ub_sanitizer():
    _make_asan()
    rts = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rts:
        pytest.skip("no clang ASan runtime under /opt/rocm")
    env = dict(os.environ, LD_PRELOAD=rts[-1], **SAN_ENV)
    r = subprocess.run([sys.executable, os.path.join(HERE, "asan_fuzz_host.py"), os.path.join(ORACLE, "_build", "librbr_hip_hostasan.so")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "ASAN HOST FUZZ OK" in r.stdout, out[-6000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
