"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol
include/mpopis.h declares; argument errors mirror the reference's error() sites.  No compute."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from mpopis_amd import build, _lib
    build.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    from mpopis_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mpopis.h")).read()
    declared = sorted(set(re.findall(r"\b(mpopis_[A-Za-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert sorted(_lib.ABI_SYMBOLS) == declared          # binding list == header
    for name in declared:
        assert hasattr(L, name), name


def test_abi_version(L):
    assert L.mpopis_abi_version() == 5      # include/mpopis.h "ABI history"


def test_null_handle_is_an_argument_error(L):
    """every entry point that takes a handle refuses NULL with MPOPIS_ERR_ARG instead of crashing (no GPU needed)"""
    assert L.mpopis_policy_call(None, None, None, None, None, None, None, None, None, None) == -1
    assert L.mpopis_policy_step(None, None, None, None, None, None, None, None) == -1
    assert L.mpopis_set_state(None, None, None, None) == -1 and L.mpopis_get_U(None, None) == -1
    assert L.mpopis_seed(None, 1) == -1 and L.mpopis_reset(None) == -1 and L.mpopis_set_overlap(None, 2) == -1


def test_config_struct_layout():
    from mpopis_amd._lib import Config
    # 10 int32 + 5 double + uint64 = 40 + 40 + 8
    assert C.sizeof(Config) == 88
    assert Config.lambda_.offset == 40 and Config.seed.offset == 80


def test_create_argument_errors_or_no_device(L):
    """Without a GPU the engine must fail loudly (no CPU fallback); bad arguments give -1 first."""
    from mpopis_amd._lib import Config
    cfg = Config()
    cfg.env_kind, cfg.num_cars, cfg.policy = 1, 1, 99
    cfg.num_samples, cfg.horizon, cfg.batch = 8, 4, 1
    h = C.c_void_p()
    assert L.mpopis_create(C.byref(cfg), C.byref(h)) == -1
    assert b"policy_type" in L.mpopis_last_error(None)
    cfg.policy = 1
    cfg.num_cars = 9
    assert L.mpopis_create(C.byref(cfg), C.byref(h)) == -1
    cfg.num_cars = 1
    import torch
    if not torch.cuda.is_available():
        rc = L.mpopis_create(C.byref(cfg), C.byref(h))
        assert rc == -4 and h.value is None
        assert b"no HIP device" in L.mpopis_last_error(None)


def test_product_never_imports_oracle():
    """The shipped package must not import, include, link or call the test oracle (no CPU fallback of any kind).
    Comments may mention it; code may not."""
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+\s*oracle|#\s*include\s*[\"<][^\n]*oracle)|\borc_[a-z_0-9]+\s*\(|libmpopis_oracle|oracle\.oracle|np_rederive",
                     re.MULTILINE)
    # the package, the C headers, the Julia shim and the developer tools: only tests/, __graft_entry__.smoke() and bench.py's
    # cpu_baseline leg may touch oracle/
    for top in ("mpopis_amd", "include", "julia", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh", ".jl")):
                    src = open(os.path.join(dp, f), errors="ignore").read()
                    m = bad.search(src)
                    assert m is None, (dp, f, m.group(0) if m else None)
    # the shared library must not depend on the oracle library either
    import subprocess
    from mpopis_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_header_is_plain_c_and_a_c_program_links(L, tmp_path):
    """include/mpopis.h is the drop-in boundary: it must be consumable by a plain C99 compiler (no C++, no torch, no HIP types in the
    signatures) and a C program must link against the shared library with nothing but -lmpopis_hip."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = os.path.join(ROOT, "tests", "abi_client.c")
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + inc, "-fsyntax-only", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    libdir = os.path.join(ROOT, "mpopis_amd", "lib")
    exe = str(tmp_path / "abi_client")
    r = subprocess.run(["gcc", "-std=c99", "-I" + inc, src, "-L" + libdir, "-lmpopis_hip", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdr = open(os.path.join(inc, "mpopis.h")).read()
    assert "torch" not in hdr and "hip/" not in hdr and "std::" not in hdr
