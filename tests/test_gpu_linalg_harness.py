"""The dense linear algebra kernels (Cholesky: LDS / register-resident / cooperative / one-workgroup global; triangular-inverse trace; Lanczos inverse square root:
one workgroup / cooperative) exercised directly, below the policy level, through the C++ harness tools/kbench_linalg.hip: sizes on both sides of
every kernel-selection threshold, batches that do and do not allow co-resident clusters."""
import os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def harness():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available to build the harness")
    from mpopis_amd import build
    build.build()                                                     # the harness links the library's object files
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_kbench_linalg.sh")], capture_output=True, text=True, timeout=600)
    exe = os.path.join(ROOT, "tools", "kbench_linalg_bin")
    assert os.path.exists(exe), out.stdout + out.stderr
    return exe


def _num(pattern, text):
    m = re.search(pattern, text)
    assert m, (pattern, text)
    return float(m.group(1))


@pytest.mark.parametrize("B,n", [(3, 16), (4, 37), (2, 100), (64, 100), (2, 128),          # k_potrf_lds
                                 (3, 129), (2, 145), (4, 200), (8, 300), (2, 333), (2, 384),   # k_potrf_coop, cooperative Lanczos from n = 160
                                 (24, 300), (48, 300), (2, 400)])                             # 48 x 6 > CUs and n = 400 (LDS too small for clusters): k_potrf_global
def test_linalg_kernels(harness, B, n):
    _check(harness, B, n, {})


@pytest.mark.parametrize("B,n", [(8, 300), (2, 256), (2, 304), (48, 300)])
def test_linalg_kernels_without_the_register_cholesky(harness, B, n):
    """n = 241 .. 304 goes to k_potrf_reg by default; the cluster / one-workgroup kernels stay behind it for n > 304 and as its reference"""
    _check(harness, B, n, {"MPOPIS_POTRF_REG": "0"})


@pytest.mark.parametrize("B,n", [(1, 300), (64, 300), (3, 250), (2, 272), (2, 304), (5, 241)])
def test_register_cholesky_has_the_bits_of_the_other_kernels(harness, B, n):
    """k_potrf_reg keeps the arithmetic of k_potrf_global / k_potrf_coop (same diagonal-block routine, same panel solve, product of the strip rows
    first and then the subtraction): the factors are bit-identical, zeros above the diagonal included (the harness poisons the output first)"""
    hashes = []
    for env in ({}, {"MPOPIS_POTRF_REG": "0"}, {"MPOPIS_POTRF_REG": "0", "MPOPIS_POTRF_G": "0"}):
        r = subprocess.run([harness, str(B), str(n)], capture_output=True, text=True, timeout=120, env=dict(os.environ, KB_POTRF_ONLY="1", **env))
        assert r.returncode == 0, r.stdout + r.stderr
        assert _num(r"potrf status min (-?\d+)", r.stdout) == 0 and _num(r"max \|upper\| = ([0-9.e+-]+)", r.stdout) == 0.0
        m = re.search(r"potrf L hash ([0-9a-f]+)", r.stdout)
        assert m, r.stdout
        hashes.append(m.group(1))
    assert hashes[0] == hashes[1] == hashes[2], hashes


def _check(harness, B, n, env):
    r = subprocess.run([harness, str(B), str(n)], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    t = r.stdout
    assert _num(r"potrf status min (-?\d+)", t) == 0
    assert _num(r"potrf max \|LL'-A\| = ([0-9.e+-]+)", t) < 1e-14              # entries of A are O(0.1)
    assert _num(r"max \|upper\| = ([0-9.e+-]+)", t) == 0.0
    assert _num(r"trtri_fro: .* rel ([0-9.e+-]+)", t) < 1e-11
    ymax = _num(r"\(\|y\| max ([0-9.e+-]+)\)", t)
    assert _num(r"lanczos coop vs single: max \|dy\| = ([0-9.e+-]+)", t) < 1e-11 * ymax
    assert _num(r"invsqrt applied twice: .* = ([0-9.e+-]+)", t) < 1e-9
    assert _num(r"status (-?\d+), tr", t) == 0
    assert _num(r"lanczos prep vs host: max rel ([0-9.e+-]+)", t) < 1e-12 and _num(r"usable (\d)", t) == 1   # spectrum bounds / quadrature nodes left by the trace launch


@pytest.mark.parametrize("B,n,decades", [(2, 20, 17), (3, 100, 16), (2, 300, 18), (9, 300, 16), (2, 301, 20)])
def test_dense_fallback_beyond_the_quadrature(harness, B, n, decades):
    """cond(A) beyond 1e14 (A = D H D, variances graded over `decades` decades): the 64-node quadrature cannot resolve the spectrum and the Lanczos launch
    hands the slot to the one-workgroup Jacobi eigen-solve (dense_invsqrt_slot) instead of reporting MPOPIS_ERR_NUMERIC -- the reference's eigen-based
    Σ^-0.5 (:580) has no conditioning limit.  Checked by identities that stay well-posed at this conditioning: y'y = b'A^-1 b (host forward substitution
    with the Cholesky factor, long double), y'A y = b'b, and tr(A^-1) against the triangular inverse's."""
    r = subprocess.run([harness, str(B), str(n)], capture_output=True, text=True, timeout=300, env=dict(os.environ, KB_GRADE=str(decades)))
    assert r.returncode == 0, r.stdout + r.stderr
    t = r.stdout
    assert _num(r"potrf status min (-?\d+)", t) == 0
    assert _num(r"msteps min (-?\d+)", t) == -1 and _num(r"msteps min -?\d+ max (-?\d+)", t) == -1           # every slot took the dense path
    assert _num(r"status min (-?\d+), y'y", t) == 0
    assert _num(r"y'y vs .* rel ([0-9.e+-]+), y'Ay", t) < 1e-9
    assert _num(r"y'Ay vs b'b rel ([0-9.e+-]+),", t) < 1e-9
    assert _num(r"triangular inverse rel ([0-9.e+-]+)", t) < 1e-9
