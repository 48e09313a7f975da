// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/link/call this.
//
// Small dense linear algebra used by the CPU restatement of the EqVIO EqF path.
// The reference uses Eigen (absent from this image); this file restates the handful of Eigen
// operations the hot path relies on, in plain C++:
//   - fixed-size matrices (Eigen::Matrix<double,R,C>)              -> orc::M<R,C>
//   - dynamic column-major matrices (Eigen::MatrixXd)              -> orc::DMat
//   - dense product, transpose                                     -> gemm()
//   - MatrixXd::inverse()  (Eigen: PartialPivLU for dynamic sizes) -> lu_inverse()
//   - MatrixXd::llt()                                              -> cholesky_lower()
//   - unsupported/MatrixFunctions .exp() (Pade + scaling/squaring) -> expm()
// Reference call sites: src/mathematical/VIO_eqf.cpp:62-135, 153-170 (products, inverse, exp),
// src/mathematical/Geometry.cpp:47 (llt).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- fixed-size matrices
template <int R, int C> struct M {
    double a[R][C];
    static M Zero() {
        M r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[i][j] = 0.0;
        return r;
    }
    static M Identity() {
        M r = Zero();
        for (int i = 0; i < (R < C ? R : C); ++i)
            r.a[i][i] = 1.0;
        return r;
    }
    double& operator()(int i, int j) { return a[i][j]; }
    const double& operator()(int i, int j) const { return a[i][j]; }
    double& operator()(int i) {
        static_assert(C == 1, "vector access");
        return a[i][0];
    }
    const double& operator()(int i) const {
        static_assert(C == 1, "vector access");
        return a[i][0];
    }
    M<C, R> T() const {
        M<C, R> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[j][i] = a[i][j];
        return r;
    }
    double squaredNorm() const {
        double s = 0;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                s += a[i][j] * a[i][j];
        return s;
    }
    double norm() const { return std::sqrt(squaredNorm()); }
    M normalized() const { return (*this) * (1.0 / norm()); }
    M operator*(double s) const {
        M r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[i][j] = a[i][j] * s;
        return r;
    }
    M operator/(double s) const {
        M r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[i][j] = a[i][j] / s;
        return r;
    }
    M operator+(const M& o) const {
        M r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[i][j] = a[i][j] + o.a[i][j];
        return r;
    }
    M operator-(const M& o) const {
        M r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                r.a[i][j] = a[i][j] - o.a[i][j];
        return r;
    }
    M operator-() const { return (*this) * -1.0; }
    template <int BR, int BC> M<BR, BC> block(int r0, int c0) const {
        M<BR, BC> r;
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j)
                r.a[i][j] = a[r0 + i][c0 + j];
        return r;
    }
    template <int BR, int BC> void setBlock(int r0, int c0, const M<BR, BC>& b) {
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j)
                a[r0 + i][c0 + j] = b.a[i][j];
    }
    bool hasNaN() const {
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j)
                if (std::isnan(a[i][j]))
                    return true;
        return false;
    }
};
template <int R, int C> M<R, C> operator*(double s, const M<R, C>& m) { return m * s; }
template <int R, int K, int C> M<R, C> operator*(const M<R, K>& A, const M<K, C>& B) {
    M<R, C> r;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k)
                s += A.a[i][k] * B.a[k][j];
            r.a[i][j] = s;
        }
    return r;
}
using Vec2 = M<2, 1>;
using Vec3 = M<3, 1>;
using Vec4 = M<4, 1>;
using Vec6 = M<6, 1>;
using Mat3 = M<3, 3>;
using Mat6 = M<6, 6>;

inline Vec3 vec3(double x, double y, double z) {
    Vec3 v;
    v(0) = x;
    v(1) = y;
    v(2) = z;
    return v;
}
inline Vec2 vec2(double x, double y) {
    Vec2 v;
    v(0) = x;
    v(1) = y;
    return v;
}
template <int N> double dot(const M<N, 1>& a, const M<N, 1>& b) {
    double s = 0;
    for (int i = 0; i < N; ++i)
        s += a(i) * b(i);
    return s;
}
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return vec3(a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0));
}
inline Mat3 skew(const Vec3& v) {
    Mat3 S = Mat3::Zero();
    S(0, 1) = -v(2);
    S(0, 2) = v(1);
    S(1, 0) = v(2);
    S(1, 2) = -v(0);
    S(2, 0) = -v(1);
    S(2, 1) = v(0);
    return S;
}
inline Mat3 inverse3(const Mat3& m) {
    // cofactor inverse (Eigen uses the same closed form for fixed 3x3)
    Mat3 r;
    const double c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
    const double c01 = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2);
    const double c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
    const double det = m(0, 0) * c00 + m(0, 1) * c01 + m(0, 2) * c02;
    const double id = 1.0 / det;
    r(0, 0) = c00 * id;
    r(1, 0) = c01 * id;
    r(2, 0) = c02 * id;
    r(0, 1) = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * id;
    r(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * id;
    r(2, 1) = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * id;
    r(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * id;
    r(1, 2) = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * id;
    r(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * id;
    return r;
}
inline M<2, 2> inverse2(const M<2, 2>& m) {
    const double det = m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0);
    M<2, 2> r;
    r(0, 0) = m(1, 1) / det;
    r(0, 1) = -m(0, 1) / det;
    r(1, 0) = -m(1, 0) / det;
    r(1, 1) = m(0, 0) / det;
    return r;
}

// ---------------------------------------------------------------- dynamic matrices (column-major)
struct DMat {
    int r = 0, c = 0;
    std::vector<double> d;
    DMat() = default;
    DMat(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
    static DMat Zero(int rows, int cols) { return DMat(rows, cols); }
    static DMat Identity(int rows, int cols) {
        DMat m(rows, cols);
        for (int i = 0; i < std::min(rows, cols); ++i)
            m(i, i) = 1.0;
        return m;
    }
    double& operator()(int i, int j) { return d[(size_t)j * r + i]; }
    const double& operator()(int i, int j) const { return d[(size_t)j * r + i]; }
    int rows() const { return r; }
    int cols() const { return c; }
    DMat T() const {
        DMat t(c, r);
        for (int j = 0; j < c; ++j)
            for (int i = 0; i < r; ++i)
                t(j, i) = (*this)(i, j);
        return t;
    }
    bool hasNaN() const {
        for (double x : d)
            if (std::isnan(x))
                return true;
        return false;
    }
    double frobenius() const {
        double s = 0;
        for (double x : d)
            s += x * x;
        return std::sqrt(s);
    }
    template <int BR, int BC> M<BR, BC> block(int r0, int c0) const {
        M<BR, BC> b;
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j)
                b(i, j) = (*this)(r0 + i, c0 + j);
        return b;
    }
    template <int BR, int BC> void setBlock(int r0, int c0, const M<BR, BC>& b) {
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j)
                (*this)(r0 + i, c0 + j) = b(i, j);
    }
};
using DVec = std::vector<double>;

inline DMat operator+(const DMat& A, const DMat& B) {
    assert(A.r == B.r && A.c == B.c);
    DMat C(A.r, A.c);
    for (size_t i = 0; i < A.d.size(); ++i)
        C.d[i] = A.d[i] + B.d[i];
    return C;
}
inline DMat operator-(const DMat& A, const DMat& B) {
    assert(A.r == B.r && A.c == B.c);
    DMat C(A.r, A.c);
    for (size_t i = 0; i < A.d.size(); ++i)
        C.d[i] = A.d[i] - B.d[i];
    return C;
}
inline DMat operator*(const DMat& A, double s) {
    DMat C(A.r, A.c);
    for (size_t i = 0; i < A.d.size(); ++i)
        C.d[i] = A.d[i] * s;
    return C;
}
inline DMat operator*(double s, const DMat& A) { return A * s; }

// C = A * B. Column-major, cache-blocked (kc x nc panels), inner loop is a contiguous axpy over a
// column of A so the compiler vectorises it. Single-threaded like the reference's Eigen products
// (no OpenMP in the reference's CMakeLists.txt).
inline DMat gemm(const DMat& A, const DMat& B) {
    assert(A.c == B.r);
    const int m = A.r, k = A.c, n = B.c;
    DMat C(m, n);
    const int KC = 256, MC = 256;
    for (int k0 = 0; k0 < k; k0 += KC) {
        const int k1 = std::min(k, k0 + KC);
        for (int i0 = 0; i0 < m; i0 += MC) {
            const int i1 = std::min(m, i0 + MC);
            for (int j = 0; j < n; ++j) {
                double* __restrict cj = &C.d[(size_t)j * m];
                int p = k0;
                for (; p + 3 < k1; p += 4) {
                    const double b0 = B(p, j), b1 = B(p + 1, j), b2 = B(p + 2, j), b3 = B(p + 3, j);
                    const double* __restrict a0 = &A.d[(size_t)p * m];
                    const double* __restrict a1 = a0 + m;
                    const double* __restrict a2 = a1 + m;
                    const double* __restrict a3 = a2 + m;
                    for (int i = i0; i < i1; ++i)
                        cj[i] += a0[i] * b0 + a1[i] * b1 + a2[i] * b2 + a3[i] * b3;
                }
                for (; p < k1; ++p) {
                    const double b0 = B(p, j);
                    const double* __restrict a0 = &A.d[(size_t)p * m];
                    for (int i = i0; i < i1; ++i)
                        cj[i] += a0[i] * b0;
                }
            }
        }
    }
    return C;
}
inline DMat operator*(const DMat& A, const DMat& B) { return gemm(A, B); }
inline DVec matvec(const DMat& A, const DVec& x) {
    assert(A.c == (int)x.size());
    DVec y(A.r, 0.0);
    for (int j = 0; j < A.c; ++j) {
        const double xj = x[j];
        const double* a = &A.d[(size_t)j * A.r];
        for (int i = 0; i < A.r; ++i)
            y[i] += a[i] * xj;
    }
    return y;
}

// Inverse via LU with partial pivoting (what Eigen's dynamic MatrixXd::inverse() does).
inline DMat lu_inverse(const DMat& Ain) {
    assert(Ain.r == Ain.c);
    const int n = Ain.r;
    DMat LU = Ain;
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(LU(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(LU(i, k)) > best) {
                best = std::fabs(LU(i, k));
                p = i;
            }
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j)
                std::swap(LU(k, j), LU(p, j));
        const double pivv = LU(k, k);
        if (pivv != 0.0) {
            const double ip = 1.0 / pivv;
            for (int i = k + 1; i < n; ++i)
                LU(i, k) *= ip;
        }
        for (int j = k + 1; j < n; ++j) {
            const double ukj = LU(k, j);
            if (ukj == 0.0)
                continue;
            double* col = &LU.d[(size_t)j * n];
            const double* lk = &LU.d[(size_t)k * n];
            for (int i = k + 1; i < n; ++i)
                col[i] -= lk[i] * ukj;
        }
    }
    // Solve LU X = P I column by column.
    DMat X = DMat::Identity(n, n);
    for (int k = 0; k < n; ++k)
        if (piv[k] != k)
            for (int j = 0; j < n; ++j)
                std::swap(X(k, j), X(piv[k], j));
    for (int j = 0; j < n; ++j) {
        double* x = &X.d[(size_t)j * n];
        for (int k = 0; k < n; ++k) {
            const double xk = x[k];
            if (xk == 0.0)
                continue;
            const double* lk = &LU.d[(size_t)k * n];
            for (int i = k + 1; i < n; ++i)
                x[i] -= lk[i] * xk;
        }
        for (int k = n - 1; k >= 0; --k) {
            x[k] /= LU(k, k);
            const double xk = x[k];
            const double* uk = &LU.d[(size_t)k * n];
            for (int i = 0; i < k; ++i)
                x[i] -= uk[i] * xk;
        }
    }
    return X;
}

// Lower Cholesky factor (Eigen LLT). Returns false if a pivot is not positive.
inline bool cholesky_lower(const DMat& A, DMat& L) {
    assert(A.r == A.c);
    const int n = A.r;
    L = DMat(n, n);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i)
            L(i, j) = A(i, j);
    for (int j = 0; j < n; ++j) {
        double* lj = &L.d[(size_t)j * n];
        for (int k = 0; k < j; ++k) {
            const double ljk = L(j, k);
            const double* lk = &L.d[(size_t)k * n];
            for (int i = j; i < n; ++i)
                lj[i] -= lk[i] * ljk;
        }
        if (!(lj[j] > 0.0))
            return false;
        const double dj = std::sqrt(lj[j]);
        lj[j] = dj;
        const double id = 1.0 / dj;
        for (int i = j + 1; i < n; ++i)
            lj[i] *= id;
    }
    return true;
}
// X <- X * L^{-T}  (right-side solve with the transposed lower factor), X is (rows x n)
inline void trsm_right_lower_trans(const DMat& L, DMat& X) {
    const int n = L.r, rows = X.r;
    assert(X.c == n);
    for (int j = 0; j < n; ++j) {
        double* xj = &X.d[(size_t)j * rows];
        for (int k = 0; k < j; ++k) {
            const double ljk = L(j, k);
            const double* xk = &X.d[(size_t)k * rows];
            for (int i = 0; i < rows; ++i)
                xj[i] -= xk[i] * ljk;
        }
        const double id = 1.0 / L(j, j);
        for (int i = 0; i < rows; ++i)
            xj[i] *= id;
    }
}
// X <- X * L^{-1}
inline void trsm_right_lower(const DMat& L, DMat& X) {
    const int n = L.r, rows = X.r;
    assert(X.c == n);
    for (int j = n - 1; j >= 0; --j) {
        double* xj = &X.d[(size_t)j * rows];
        const double id = 1.0 / L(j, j);
        for (int i = 0; i < rows; ++i)
            xj[i] *= id;
        for (int k = 0; k < j; ++k) {
            const double ljk = L(j, k);
            double* xk = &X.d[(size_t)k * rows];
            for (int i = 0; i < rows; ++i)
                xk[i] -= xj[i] * ljk;
        }
    }
}

inline double norm1(const DMat& A) {
    double best = 0;
    for (int j = 0; j < A.c; ++j) {
        double s = 0;
        for (int i = 0; i < A.r; ++i)
            s += std::fabs(A(i, j));
        best = std::max(best, s);
    }
    return best;
}
// Solve (V - U) X = (V + U) by LU with partial pivoting (general solve, used by expm).
inline DMat lu_solve(const DMat& A, const DMat& B) { return gemm(lu_inverse(A), B); }

// Matrix exponential: Pade approximants of degree 3/5/7/9/13 with scaling and squaring
// (Higham 2005) — the algorithm behind Eigen's unsupported MatrixBase::exp() used at
// src/mathematical/VIO_eqf.cpp:85.
inline DMat expm(const DMat& A) {
    assert(A.r == A.c);
    const int n = A.r;
    const DMat I = DMat::Identity(n, n);
    const double l1 = norm1(A);
    DMat U, V;
    int squarings = 0;
    auto pade3 = [&](const DMat& A_) {
        const double b[] = {120., 60., 12., 1.};
        const DMat A2 = A_ * A_;
        const DMat tmp = b[3] * A2 + b[1] * I;
        U = A_ * tmp;
        V = b[2] * A2 + b[0] * I;
    };
    auto pade5 = [&](const DMat& A_) {
        const double b[] = {30240., 15120., 3360., 420., 30., 1.};
        const DMat A2 = A_ * A_;
        const DMat A4 = A2 * A2;
        const DMat tmp = b[5] * A4 + b[3] * A2 + b[1] * I;
        U = A_ * tmp;
        V = b[4] * A4 + b[2] * A2 + b[0] * I;
    };
    auto pade7 = [&](const DMat& A_) {
        const double b[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        const DMat A2 = A_ * A_;
        const DMat A4 = A2 * A2;
        const DMat A6 = A4 * A2;
        const DMat tmp = b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
        U = A_ * tmp;
        V = b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
    };
    auto pade9 = [&](const DMat& A_) {
        const double b[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                            2162160.,     110880.,     3960.,       90.,        1.};
        const DMat A2 = A_ * A_;
        const DMat A4 = A2 * A2;
        const DMat A6 = A4 * A2;
        const DMat A8 = A6 * A2;
        const DMat tmp = b[9] * A8 + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
        U = A_ * tmp;
        V = b[8] * A8 + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
    };
    auto pade13 = [&](const DMat& A_) {
        const double b[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                            129060195264000.,   10559470521600.,    670442572800.,     33522128640.,
                            1323241920.,        40840800.,          960960.,           16380.,
                            182.,               1.};
        const DMat A2 = A_ * A_;
        const DMat A4 = A2 * A2;
        const DMat A6 = A4 * A2;
        V = b[13] * A6 + b[11] * A4 + b[9] * A2;
        DMat tmp = A6 * V;
        tmp = tmp + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I;
        U = A_ * tmp;
        tmp = b[12] * A6 + b[10] * A4 + b[8] * A2;
        V = A6 * tmp;
        V = V + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I;
    };
    if (l1 < 1.495585217958292e-002) {
        pade3(A);
    } else if (l1 < 2.539398330063230e-001) {
        pade5(A);
    } else if (l1 < 9.504178996162932e-001) {
        pade7(A);
    } else if (l1 < 2.097847961257068e+000) {
        pade9(A);
    } else {
        const double maxnorm = 5.371920351148152;
        int e = 0;
        std::frexp(l1 / maxnorm, &e);
        squarings = std::max(0, e);
        const DMat As = A * std::ldexp(1.0, -squarings);
        pade13(As);
    }
    DMat R = lu_solve(V - U, V + U);
    for (int i = 0; i < squarings; ++i)
        R = R * R;
    return R;
}

} // namespace orc
