// Host-side design of the product-form propagator polynomial.
//
// exp(-i*rho*x), x in [-1,1], is approximated by its truncated Chebyshev expansion
//     p(x) = J_0(rho) + 2 * sum_{k=1..m} (-i)^k J_k(rho) T_k(x)
// and applied in FACTORED form  p(x) = p(0) * prod_f (1 - x/z_f)  (z_f = roots of p, sorted by
// decreasing modulus, which keeps every partial product O(1) on [-1,1]).  One factor is one
// matrix-free "y = gamma*x + beta*H*x" pass: read the vector once, write it once, no accumulator
// vector and no global reduction — which is what makes the propagator a pure streaming workload.
//
// This replaces the Lanczos iteration of the reference's KRYLOV_SE solver (pyqtorch.sesolve, call
// site pulser_diff/backend.py:488-494): same map psi -> exp(-i*H*dt) psi, to `tol`.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace rydiff {

using cld = std::complex<long double>;
using cdd = std::complex<double>;

struct PolyDesign {
    double rho = 0.0;
    double tol = 0.0;
    int degree = 0;
    std::vector<cdd> roots;  // sorted by |z| descending
    cdd p0{1.0, 0.0};        // p(0)
    double max_err = 0.0;    // measured on a Chebyshev grid with the product form in double
};

// Bessel J_0..J_kmax(rho) by Miller's backward recurrence (long double).
inline std::vector<long double> bessel_j_upto(long double rho, int kmax) {
    std::vector<long double> j(kmax + 1, 0.0L);
    if (rho == 0.0L) {
        j[0] = 1.0L;
        return j;
    }
    int start = kmax + 40 + static_cast<int>(2.0L * rho);
    if (start % 2) ++start;
    long double jp1 = 0.0L, jc = 1e-300L, sum = 0.0L;
    for (int k = start; k >= 0; --k) {
        // J_{k-1} = (2k/rho) J_k - J_{k+1}; at loop entry jc = J_k (unnormalised)
        if (k <= kmax) j[k] = jc;
        if (k % 2 == 0) sum += (k == 0 ? jc : 2.0L * jc);
        long double jm1 = (k > 0) ? (2.0L * k / rho) * jc - jp1 : 0.0L;
        jp1 = jc;
        jc = jm1;
        if (fabsl(jp1) > 1e250L) {  // rescale
            const long double s = 1e-250L;
            jp1 *= s;
            jc *= s;
            sum *= s;
            for (int q = k; q <= kmax; ++q) j[q] *= s;
        }
    }
    for (auto& v : j) v /= sum;
    return j;
}

inline void cheb_eval(const std::vector<cld>& a, cld z, cld& p, cld& dp) {
    // forward recurrences T_{k+1} = 2 z T_k - T_{k-1};  T'_{k+1} = 2 T_k + 2 z T'_k - T'_{k-1}
    cld t0 = 1.0L, t1 = z, d0 = 0.0L, d1 = 1.0L;
    p = a[0];
    dp = 0.0L;
    if (a.size() > 1) {
        p += a[1] * t1;
        dp += a[1] * d1;
    }
    for (size_t k = 2; k < a.size(); ++k) {
        cld t2 = 2.0L * z * t1 - t0;
        cld d2 = 2.0L * t1 + 2.0L * z * d1 - d0;
        p += a[k] * t2;
        dp += a[k] * d2;
        t0 = t1;
        t1 = t2;
        d0 = d1;
        d1 = d2;
    }
}

inline bool aberth_roots(const std::vector<cld>& a, long double radius, std::vector<cld>& z) {
    const int m = static_cast<int>(a.size()) - 1;
    z.resize(m);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < m; ++i) {
        long double th = 2.0L * pi * (i + 0.37L) / m;
        z[i] = cld(radius * cosl(th), radius * sinl(th) * 1.07L);
    }
    long double last = 1.0L;
    for (int it = 0; it < 300; ++it) {
        long double maxstep = 0.0L;
        for (int i = 0; i < m; ++i) {
            cld p, dp;
            cheb_eval(a, z[i], p, dp);
            if (std::abs(p) == 0.0L) continue;
            cld w = p / dp;
            cld s = 0.0L;
            for (int j = 0; j < m; ++j)
                if (j != i) s += 1.0L / (z[i] - z[j]);
            cld step = w / (1.0L - w * s);
            z[i] -= step;
            maxstep = std::max(maxstep, std::abs(step) / std::max(1.0L, std::abs(z[i])));
        }
        // converged, or stagnating at the floor set by the conditioning of the root problem
        if (maxstep < 1e-17L || (maxstep < 1e-12L && maxstep > 0.25L * last)) return true;
        last = maxstep;
    }
    // conditioning of the root problem limits the attainable step size; the caller validates the product form
    return last < 1e-11L;
}

inline int degree_for(double rho, double tol) {
    const int kmax = 400;
    auto j = bessel_j_upto(rho, kmax);
    // smallest m with 2*sum_{k>m} |J_k| < tol
    long double tail = 0.0L;
    int m = kmax;
    while (m > 1) {
        long double next = tail + 2.0L * fabsl(j[m]);
        if (next >= tol) break;
        tail = next;
        --m;
    }
    return std::max(m, 1);
}

inline PolyDesign design_polynomial(double rho, double tol, int max_degree = 120) {
    PolyDesign d;
    d.rho = rho;
    d.tol = tol;
    if (!(tol > 1e-15)) tol = 1e-15;
    int m = std::min(degree_for(rho, tol), max_degree);
    for (int attempt = 0; attempt < 4; ++attempt) {
        auto j = bessel_j_upto(rho, m);
        std::vector<cld> a(m + 1);
        const cld mi(0.0L, -1.0L);
        cld pw = 1.0L;
        for (int k = 0; k <= m; ++k) {
            a[k] = (k == 0 ? 1.0L : 2.0L) * pw * j[k];
            pw *= mi;
        }
        std::vector<cld> z;
        bool ok = false;
        const long double base = 0.6L * m / std::max<long double>(rho, 1e-12L) + 1.0L;
        for (long double scale : {1.0L, 1.7L, 0.6L, 3.0L}) {
            if (aberth_roots(a, base * scale, z)) {
                ok = true;
                break;
            }
        }
        cld p0, dp0;
        cheb_eval(a, cld(0.0L), p0, dp0);
        if (ok) {
            std::sort(z.begin(), z.end(), [](const cld& u, const cld& v) { return std::abs(u) > std::abs(v); });
            d.degree = m;
            d.roots.resize(m);
            for (int i = 0; i < m; ++i) d.roots[i] = cdd(static_cast<double>(z[i].real()), static_cast<double>(z[i].imag()));
            d.p0 = cdd(static_cast<double>(p0.real()), static_cast<double>(p0.imag()));
            // measure the product form in double arithmetic, exactly as the device applies it
            double err = 0.0;
            const int npts = 257;
            for (int i = 0; i < npts; ++i) {
                double x = std::cos(M_PI * i / (npts - 1));
                cdd p = d.p0;
                for (int f = 0; f < m; ++f) p *= (1.0 - x / d.roots[f]);
                cdd ex(std::cos(rho * x), -std::sin(rho * x));
                err = std::max(err, std::abs(p - ex));
            }
            d.max_err = err;
            if (err < 40.0 * tol + 4e-14 || m >= max_degree) return d;
        }
        m = std::min(m + 2, max_degree);
    }
    return d;
}

}  // namespace rydiff
