// kbench_linalg.hip -- standalone timing of the one-workgroup-per-slot linear-algebra kernels at CMA sizes (dev tool, not shipped).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench_linalg.hip mpopis_amd/lib/obj/kernels_linalg.o mpopis_amd/lib/obj/kernels_invsqrt.o -o tools/kbench_linalg_bin
#include "../mpopis_amd/csrc/engine.h"
#include "../mpopis_amd/csrc/invsqrt_quad.h"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <cstring>
using namespace mpopis;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class F> float timeit(F f, int reps, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
#ifdef POTRF_PROF
namespace mpopis { void debug_read_rprof(unsigned long long* out); void debug_read_prof(unsigned long long* out); void debug_read_lprof(unsigned long long* out); }
#endif
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 300;
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t nn = (size_t)n * n;
    std::mt19937_64 rng(1); std::normal_distribution<double> nd;
    // CMA-like covariance: two-eigenvalue block diagonal + a few rank-one terms + a constant added to every entry
    std::vector<double> A(nn * B, 0.0), bv((size_t)B * n);
    for (int b = 0; b < B; ++b) {
        double* a = A.data() + b * nn;
        for (int i = 0; i < n; ++i) a[i + (size_t)i * n] = (i & 1) ? 0.1 : 0.0625;
        for (int t = 0; t < 6; ++t) {
            std::vector<double> p(n); for (auto& v : p) v = nd(rng) * 0.05;
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) a[i + (size_t)j * n] = 0.99 * a[i + (size_t)j * n] + 1e-3 * p[i] * p[j] + 2e-5;
        }
        for (int i = 0; i < n; ++i) bv[(size_t)b * n + i] = nd(rng);
        if (const char* gr = getenv("KB_GRADE")) {          // A <- D A D, D graded over KB_GRADE decades in the variances: cond(A) ~ 10^KB_GRADE, still numerically SPD (scaled-diagonally-dominant)
            const double dec = atof(gr);
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i)
                a[i + (size_t)j * n] *= pow(10.0, -0.5 * dec * i / (n - 1.0)) * pow(10.0, -0.5 * dec * j / (n - 1.0));
        }
    }
    double *dA, *dL, *db, *dpart, *dV, *dy, *dfro; int *dstatus, *dact, *dm;
    CK(hipMalloc(&dA, (nn * B + kInvsqrtPadDoubles) * 8)); CK(hipMalloc(&dL, nn * B * 8)); CK(hipMalloc(&db, (size_t)B * n * 8));
    CK(hipMalloc(&dpart, (size_t)B * ((n + 15) / 16) * 8)); const int lanG = invsqrt_coop_groups(B, n); CK(hipMalloc(&dV, invsqrt_workspace_doubles(B, n, lanG) * 8));
    unsigned long long* dlx; CK(hipMalloc(&dlx, invsqrt_coop_words(B, n) * 8)); CK(hipMemset(dlx, 0, invsqrt_coop_words(B, n) * 8)); unsigned long long lep = 0;
    CK(hipMalloc(&dy, (size_t)B * n * 8)); CK(hipMalloc(&dfro, B * 8)); CK(hipMalloc(&dstatus, B * 4)); CK(hipMalloc(&dact, B * 4)); CK(hipMalloc(&dm, B * 4));
    CK(hipMemcpy(dA, A.data(), nn * B * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bv.data(), (size_t)B * n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dstatus, 0, B * 4));
    std::vector<int> ones(B, 1); CK(hipMemcpy(dact, ones.data(), B * 4, hipMemcpyHostToDevice));
    printf("B=%d n=%d\n", B, n);
    unsigned long long* dflags; unsigned long long epoch = 0;
    CK(hipMalloc(&dflags, potrf_coop_flag_words(B, n) * 8)); CK(hipMemset(dflags, 0, potrf_coop_flag_words(B, n) * 8));
    int* dredo; CK(hipMalloc(&dredo, (2 * B + 1) * 4)); CK(hipMemset(dredo, 0, (2 * B + 1) * 4));
    CoopCtx pc; pc.flags = dflags; pc.epoch = &epoch; pc.redo = dredo; pc.timeouts = dredo + 2 * B;
    CoopCtx lc; lc.flags = dlx; lc.epoch = &lep; lc.redo = dredo + B; lc.timeouts = dredo + 2 * B;
    if (getenv("KB_NO_COOP")) { pc = CoopCtx(); lc = CoopCtx(); }
    CK(hipMemset(dL, 0xff, nn * B * 8));                 // poison: the kernels must write the zeros above the diagonal themselves
    printf("potrf                 %8.1f us\n", timeit([&] { launch_potrf(dA, nn, dL, B, n, nullptr, dstatus, dact, s, pc); }, 20, s));
#ifdef POTRF_PROF
    if (getenv("KB_RPROF")) {
        std::vector<unsigned long long> pr(8 * 32 * 6); mpopis::debug_read_rprof(pr.data());
        const int npan = (n + 15) / 16;
        unsigned long long t00 = ~0ull; for (auto v : pr) if (v && v < t00) t00 = v;
        printf("   register Cholesky, per stage j and wave: start solved zeros barrierX diag updates  [us, 100 MHz clock]\n");
        for (int w = 0; w < 8; ++w) { const unsigned long long* q = &pr[(w * 32 + 31) * 6]; printf("   load w=%d: start %7.2f loads-consumed %7.2f barrier %7.2f\n", w, (q[0]-t00)*0.01, (q[1]-t00)*0.01, (q[2]-t00)*0.01); }
        for (int j = 0; j < npan; ++j) for (int w = 0; w < 8; ++w) { const unsigned long long* q = &pr[(w * 32 + j) * 6];
            printf("   j=%2d w=%d: %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", j, w, (q[0]-t00)*0.01, (q[1]-t00)*0.01, (q[2]-t00)*0.01, (q[3]-t00)*0.01, q[4] ? (q[4]-t00)*0.01 : 0.0, q[5] ? (q[5]-t00)*0.01 : 0.0); }
    } else {
        std::vector<unsigned long long> pr(8 * 32 * 6); mpopis::debug_read_prof(pr.data());
        const int G = getenv("MPOPIS_POTRF_G") ? atoi(getenv("MPOPIS_POTRF_G")) : 6, npan = (n + 15) / 16;
        unsigned long long t00 = ~0ull; for (auto v : pr) if (v && v < t00) t00 = v;
        printf("   step owner(j+1): start recv upd diag publish end   [us, 100 MHz clock]\n");
        for (int j = 0; j + 1 < npan; ++j) { const int g = (j + 1) % G; const unsigned long long* q = &pr[(g * 32 + j) * 6];
            printf("   j=%2d g=%d: %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", j, g, (q[0]-t00)*0.01, (q[1]-t00)*0.01, (q[2]-t00)*0.01, (q[5]-t00)*0.01, (q[3]-t00)*0.01, (q[4]-t00)*0.01); }
    }
#endif
    { std::vector<int> st0(B); CK(hipMemcpy(st0.data(), dstatus, B * 4, hipMemcpyDeviceToHost)); int mn = 0; for (int v : st0) mn = v < mn ? v : mn; printf("   potrf status min %d\n", mn); }
    {
        std::vector<double> Lall(nn * B); CK(hipMemcpy(Lall.data(), dL, nn * B * 8, hipMemcpyDeviceToHost));
        double err = 0, up = 0;
        for (int bb = 0; bb < B; bb += (B > 4 ? B / 4 : 1)) {
            const double* L = Lall.data() + bb * nn; const double* Ab = A.data() + bb * nn;
            for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = 0; for (int k = 0; k <= j; ++k) v += L[i + (size_t)k * n] * L[j + (size_t)k * n]; err = fmax(err, fabs(v - Ab[i + (size_t)j * n])); }
            for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) up = fmax(up, fabs(L[i + (size_t)j * n]));
        }
        printf("   potrf max |LL'-A| = %.3e, max |upper| = %.1e\n", err, up);
        unsigned long long h = 1469598103934665603ull;
        for (size_t e = 0; e < Lall.size(); ++e) { unsigned long long w; memcpy(&w, &Lall[e], 8); h = (h ^ w) * 1099511628211ull; }
        printf("   potrf L hash %016llx\n", h);
    }
    if (getenv("KB_POTRF_ONLY")) return 0;
    double* ddinv; CK(hipMalloc(&ddinv, trtri_dinv_doubles(B, n) * 8));
    double* dprep; CK(hipMalloc(&dprep, lanczos_prep_doubles(B) * 8));
    unsigned long long* dcnt; CK(hipMalloc(&dcnt, B * 16)); CK(hipMemset(dcnt, 0, B * 16));
    printf("trtri_fro             %8.1f us\n", timeit([&] { launch_trtri_fro(dL, nn, dpart, B, n, dact, s, ddinv, true, dA, nullptr, dprep, dcnt); }, 20, s));
    if (getenv("KB_LANCZOS_ONCE")) {
        launch_trtri_fro(dL, nn, dpart, B, n, dact, s, ddinv, true, dA, nullptr, dprep, dcnt);
        for (int w = 0; w < 3; ++w) launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, lanG, lc);     // warm
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, s);
        launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, lanG, lc);
        hipEventRecord(e1, s); CK(hipStreamSynchronize(s)); float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<int> m1(B), st1(B); CK(hipMemcpy(m1.data(), dm, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(st1.data(), dstatus, B * 4, hipMemcpyDeviceToHost));
        printf("one lanczos launch (G = %d): %.3f ms, m[0] = %d, status[0] = %d\n", lanG, ms, m1[0], st1[0]);
#ifdef POTRF_PROF
        { std::vector<unsigned long long> lp(64 * 8); mpopis::debug_read_lprof(lp.data()); const unsigned long long t00 = lp[0];
          printf("   step: start | matvec+publish | exchange | cgs2 | norm+stop | (us since kernel's first step)\n");
          for (int g = 0; g < lanG && g < 8; ++g) printf("   workgroup %d: entry %7.2f slab loaded %7.2f loop %7.2f first mat-vec published %7.2f first exchange done %7.2f\n", g, (double)(long long)(lp[(48+g)*8]-t00)*0.01, (double)(long long)(lp[(48+g)*8+1]-t00)*0.01, (double)(long long)(lp[(48+g)*8+2]-t00)*0.01, (double)(long long)(lp[(48+g)*8+3]-t00)*0.01, (double)(long long)(lp[(48+g)*8+4]-t00)*0.01);
          for (int j = 0; j <= m1[0] && j < 48; ++j) printf("   j=%2d %7.2f %7.2f %7.2f %7.2f %7.2f\n", j, (lp[j*8]-t00)*0.01, (lp[j*8+1]-t00)*0.01, (lp[j*8+2]-t00)*0.01, (lp[j*8+3]-t00)*0.01, (lp[j*8+4]-t00)*0.01); }
#endif
        std::vector<unsigned long long> xb(4 * (n + lanG)); CK(hipMemcpy(xb.data(), dlx, xb.size() * 8, hipMemcpyDeviceToHost));
        for (int i : {0, 1, 37, 38, 39, 150, 299, 300, 301, 307}) if (i < n + lanG) printf("   x[%d] = %016llx %016llx | parity1 %016llx %016llx\n", i, xb[2 * i], xb[2 * i + 1], xb[2 * (n + lanG) + 2 * i], xb[2 * (n + lanG) + 2 * i + 1]);
        return 0;
    }
    if (getenv("KB_GRADE")) {
        // beyond the quadrature's range the Lanczos launch hands the slot to the dense (Jacobi) fall-back: msteps = -1, status stays 0; applied twice it must invert A,
        // and its trace must be the one the triangular inverse gives
        launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, lanG, lc);
        CK(hipStreamSynchronize(s));
        std::vector<int> m1(B), st1(B); CK(hipMemcpy(m1.data(), dm, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(st1.data(), dstatus, B * 4, hipMemcpyDeviceToHost));
        std::vector<double> fro1(B); CK(hipMemcpy(fro1.data(), dfro, B * 8, hipMemcpyDeviceToHost));
        // Two identities of y = A^-1/2 b that stay well-posed at cond(A) ~ 1e17+ (||A y2 - b|| / ||b|| does not: it weighs the eigenvector entries that
        // are 1e-17 of the vector's norm):  y'y = b'A^-1 b = ||L^-1 b||^2 (host forward substitution with the device factor) and y'A y = b'b.
        std::vector<double> y1((size_t)B * n), Lh(nn * B); CK(hipMemcpy(y1.data(), dy, y1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Lh.data(), dL, nn * B * 8, hipMemcpyDeviceToHost));
        double worst = 0, worst2 = 0; int mmin = 0, mmax = -9, smin = 0;
        for (int bb = 0; bb < B; ++bb) {
            const double* L = Lh.data() + bb * nn; const double* bq = bv.data() + (size_t)bb * n; const double* yq = y1.data() + (size_t)bb * n;
            std::vector<long double> x(n);
            long double xx = 0, yy = 0, yAy = 0, bb2 = 0;
            for (int i = 0; i < n; ++i) { long double v = bq[i]; for (int k = 0; k < i; ++k) v -= (long double)L[i + (size_t)k * n] * x[k]; x[i] = v / L[i + (size_t)i * n]; xx += x[i] * x[i]; }
            for (int i = 0; i < n; ++i) { yy += (long double)yq[i] * yq[i]; bb2 += (long double)bq[i] * bq[i]; long double v = 0; for (int j = 0; j < n; ++j) v += (long double)A[bb * nn + i + (size_t)j * n] * yq[j]; yAy += v * yq[i]; }
            worst = fmax(worst, (double)fabsl(yy - xx) / (double)xx); worst2 = fmax(worst2, (double)fabsl(yAy - bb2) / (double)bb2);
            mmin = m1[bb] < mmin ? m1[bb] : mmin; mmax = m1[bb] > mmax ? m1[bb] : mmax; smin = st1[bb] < smin ? st1[bb] : smin;
        }
        std::vector<double> pall((size_t)B * ((n + 15) / 16)); CK(hipMemcpy(pall.data(), dpart, pall.size() * 8, hipMemcpyDeviceToHost));
        double trw = 0;
        for (int bb = 0; bb < B; ++bb) { double fr = 0; for (int J = 0; J < (n + 15) / 16; ++J) fr += pall[(size_t)bb * ((n + 15) / 16) + J]; trw = fmax(trw, fabs(fro1[bb] - fr) / fr); }
        printf("   dense fall-back: msteps min %d max %d, status min %d, y'y vs ||L^-1 b||^2 rel %.3e, y'Ay vs b'b rel %.3e, tr(A^-1) vs triangular inverse rel %.3e\n", mmin, mmax, smin, worst, worst2, trw);
        return 0;
    }
    printf("lanczos (G = %d)       %8.1f us\n", lanG, timeit([&] { launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, lanG, lc); }, 20, s));
    {   // y against the one-workgroup kernel
        std::vector<double> y1((size_t)B * n), y0((size_t)B * n);
        CK(hipMemcpy(y1.data(), dy, y1.size() * 8, hipMemcpyDeviceToHost));
        launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, 1, CoopCtx());
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(y0.data(), dy, y0.size() * 8, hipMemcpyDeviceToHost));
        double d = 0, nr = 0; for (size_t i = 0; i < y0.size(); ++i) { d = fmax(d, fabs(y1[i] - y0[i])); nr = fmax(nr, fabs(y0[i])); }
        printf("   lanczos coop vs single: max |dy| = %.3e (|y| max %.3e)\n", d, nr);
    }
    {   // applying the operator twice must invert A: y2 = A^-1/2 (A^-1/2 b) = A^-1 b  ->  ||A y2 - b|| / ||b||
        double* dy2; CK(hipMalloc(&dy2, (size_t)B * n * 8));
        launch_lanczos_invsqrt(dA, dprep, db, n, dV, dy, dfro, dm, B, n, dstatus, dact, s, lanG, lc);
        launch_lanczos_invsqrt(dA, dprep, dy, n, dV, dy2, dfro, dm, B, n, dstatus, dact, s, lanG, lc);
        CK(hipStreamSynchronize(s));
        std::vector<double> y2((size_t)B * n); CK(hipMemcpy(y2.data(), dy2, y2.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int bb = 0; bb < B; ++bb) {
            double r2 = 0, b2 = 0;
            for (int i = 0; i < n; ++i) { double v = 0; for (int j = 0; j < n; ++j) v += A[bb * nn + i + (size_t)j * n] * y2[(size_t)bb * n + j]; v -= bv[(size_t)bb * n + i]; r2 += v * v; b2 += bv[(size_t)bb * n + i] * bv[(size_t)bb * n + i]; }
            worst = fmax(worst, sqrt(r2 / b2));
        }
        printf("   invsqrt applied twice: max ||A y2 - b|| / ||b|| = %.3e\n", worst);
        // tr(A^-1) from the triangular inverse against the host value from L (forward substitution on the identity)
        std::vector<double> L0(nn); CK(hipMemcpy(L0.data(), dL, nn * 8, hipMemcpyDeviceToHost));
        double tr = 0; std::vector<double> x(n);
        for (int c = 0; c < n; ++c) { for (int i = 0; i < n; ++i) { double v = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) v -= L0[i + (size_t)k * n] * x[k]; x[i] = (i < c) ? 0.0 : v / L0[i + (size_t)i * n]; } for (int i = c; i < n; ++i) tr += x[i] * x[i]; }
        const int np = (n + 15) / 16; std::vector<double> pp(np); CK(hipMemcpy(pp.data(), dpart, np * 8, hipMemcpyDeviceToHost));
        double trd = 0; for (double v : pp) trd += v;
        printf("   trtri_fro: ||L^-1||_F^2 device %.12e host %.12e rel %.2e\n", trd, tr, fabs(trd - tr) / tr);
    }
    {   // what the trace launch's last workgroup left for the Lanczos kernel (lanczos_prep_slot), against the host: fro = sum of the partial traces,
        // M = max column abs sum, m = min(1 / fro, M / 2), the 64 quadrature nodes of invsqrt_quad.h for [m, M] (host libm vs device sin / asin: rounding-level apart)
        const int np = (n + 15) / 16, NP = 4 + 128;
        std::vector<double> pr((size_t)B * NP), pall((size_t)B * np);
        CK(hipMemcpy(pr.data(), dprep, pr.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(pall.data(), dpart, pall.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0; int usable = 1;
        for (int bb = 0; bb < B; ++bb) {
            double fr = 0; for (int J = 0; J < np; ++J) fr += pall[(size_t)bb * np + J];
            double M = 0; for (int c = 0; c < n; ++c) { double a = 0; for (int i = 0; i < n; ++i) a += fabs(A[bb * nn + i + (size_t)c * n]); M = fmax(M, a); }
            const double* o = &pr[(size_t)bb * NP];
            const double mlo = fmin(1.0 / fr, 0.5 * M);
            worst = fmax(worst, fmax(fabs(o[0] - fr) / fr, fmax(fabs(o[1] - M) / M, fabs(o[2] - mlo) / mlo)));
            usable &= (o[3] == 1.0);
            for (int j = 0; j < 64; ++j) {
                double sh, wt;
                if (!invsqrt_quad_node(o[2], o[1], j, 64, &sh, &wt)) { usable = 0; break; }
                worst = fmax(worst, fmax(fabs(o[4 + j] - sh) / sh, fabs(o[4 + 64 + j] - wt) / wt));
            }
        }
        printf("   lanczos prep vs host: max rel %.3e, usable %d\n", worst, usable);
    }
    std::vector<int> m(B), st(B); std::vector<double> fro(B);
    CK(hipMemcpy(m.data(), dm, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), dstatus, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(fro.data(), dfro, B * 8, hipMemcpyDeviceToHost));
    printf("   Lanczos steps m = %d %d ..., status %d, tr(A^-1) = %.6e\n", m[0], m[B > 1 ? 1 : 0], st[0], fro[0]);
    { int to = 0; CK(hipMemcpy(&to, dredo + 2 * B, 4, hipMemcpyDeviceToHost)); printf("   cooperative time-outs (slots redone by the one-workgroup kernels): %d\n", to); }
    return 0;
}
