// Small fixed-size fp64 geometry for the MI355X EqF path, usable from HIP kernels and from the host
// side of the C-ABI (HD = __host__ __device__). Plain structs of doubles: everything stays in VGPRs in
// the per-landmark kernels (no local-memory arrays indexed at run time).
//
// Conventions follow the reference's use of LiePP: unit quaternion (w,x,y,z),
// SE3 = (R,x) with 6-vectors ordered (omega, v), SOT3 = (R,a) acting as a*R*p.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace eqf {

constexpr double kGravity = 9.80665; // include/eqvio/mathematical/IMUVelocity.h:26

struct V3 {
    double x, y, z;
};
struct Qt { // unit quaternion
    double w, x, y, z;
};
struct M3 { // row-major 3x3
    double a00, a01, a02, a10, a11, a12, a20, a21, a22;
};
struct Pose { // SE3
    Qt R;
    V3 x;
};

HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
HD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
HD V3 operator*(V3 a, double s) { return V3{s * a.x, s * a.y, s * a.z}; }
HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
HD double norm2(V3 a) { return dot(a, a); }
HD double norm(V3 a) { return sqrt(dot(a, a)); }
HD V3 normalized(V3 a) { return a * (1.0 / norm(a)); }

HD M3 m3_identity() { return M3{1, 0, 0, 0, 1, 0, 0, 0, 1}; }
HD M3 operator*(const M3& A, const M3& B) {
    return M3{A.a00 * B.a00 + A.a01 * B.a10 + A.a02 * B.a20, A.a00 * B.a01 + A.a01 * B.a11 + A.a02 * B.a21,
              A.a00 * B.a02 + A.a01 * B.a12 + A.a02 * B.a22, A.a10 * B.a00 + A.a11 * B.a10 + A.a12 * B.a20,
              A.a10 * B.a01 + A.a11 * B.a11 + A.a12 * B.a21, A.a10 * B.a02 + A.a11 * B.a12 + A.a12 * B.a22,
              A.a20 * B.a00 + A.a21 * B.a10 + A.a22 * B.a20, A.a20 * B.a01 + A.a21 * B.a11 + A.a22 * B.a21,
              A.a20 * B.a02 + A.a21 * B.a12 + A.a22 * B.a22};
}
HD V3 operator*(const M3& A, V3 v) {
    return V3{A.a00 * v.x + A.a01 * v.y + A.a02 * v.z, A.a10 * v.x + A.a11 * v.y + A.a12 * v.z,
              A.a20 * v.x + A.a21 * v.y + A.a22 * v.z};
}
HD M3 operator*(double s, const M3& A) {
    return M3{s * A.a00, s * A.a01, s * A.a02, s * A.a10, s * A.a11, s * A.a12, s * A.a20, s * A.a21, s * A.a22};
}
HD M3 operator+(const M3& A, const M3& B) {
    return M3{A.a00 + B.a00, A.a01 + B.a01, A.a02 + B.a02, A.a10 + B.a10, A.a11 + B.a11,
              A.a12 + B.a12, A.a20 + B.a20, A.a21 + B.a21, A.a22 + B.a22};
}
HD M3 operator-(const M3& A, const M3& B) {
    return M3{A.a00 - B.a00, A.a01 - B.a01, A.a02 - B.a02, A.a10 - B.a10, A.a11 - B.a11,
              A.a12 - B.a12, A.a20 - B.a20, A.a21 - B.a21, A.a22 - B.a22};
}
HD M3 transpose(const M3& A) { return M3{A.a00, A.a10, A.a20, A.a01, A.a11, A.a21, A.a02, A.a12, A.a22}; }
HD M3 skew(V3 v) { return M3{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}; }
HD M3 outer(V3 a, V3 b) {
    return M3{a.x * b.x, a.x * b.y, a.x * b.z, a.y * b.x, a.y * b.y, a.y * b.z, a.z * b.x, a.z * b.y, a.z * b.z};
}
HD V3 row(const M3& A, int r) { return r == 0 ? V3{A.a00, A.a01, A.a02} : (r == 1 ? V3{A.a10, A.a11, A.a12} : V3{A.a20, A.a21, A.a22}); }
HD V3 col(const M3& A, int c) { return c == 0 ? V3{A.a00, A.a10, A.a20} : (c == 1 ? V3{A.a01, A.a11, A.a21} : V3{A.a02, A.a12, A.a22}); }
HD M3 m3_rows(V3 r0, V3 r1, V3 r2) { return M3{r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z}; }
HD M3 m3_cols(V3 c0, V3 c1, V3 c2) { return M3{c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z}; }

// ---- SO(3) as quaternion
HD Qt q_identity() { return Qt{1, 0, 0, 0}; }
HD Qt q_unit(Qt q) {
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return Qt{q.w * n, q.x * n, q.y * n, q.z * n};
}
// Composition re-normalises: with conjugate-as-inverse, B <- T^-1 A T (VIOGroup.cpp:249) would otherwise
// triple any norm error per IMU step and diverge within a few frames.
HD Qt q_mul(Qt a, Qt b) {
    return q_unit(Qt{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                     a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x});
}
HD Qt q_inv(Qt a) { return Qt{a.w, -a.x, -a.y, -a.z}; }
HD V3 q_rot(Qt q, V3 v) { // v + 2w(u x v) + 2 u x (u x v)
    const V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
HD M3 q_mat(Qt q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return M3{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
}
HD Qt so3_exp(V3 om) {
    const double th = norm(om);
    const double half = 0.5 * th;
    double s;
    if (th < 1e-6) {
        const double t2 = th * th;
        s = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0;
    } else {
        s = sin(half) / th;
    }
    return Qt{cos(half), s * om.x, s * om.y, s * om.z};
}
// minimal rotation with R a = b  (Eigen Quaternion::setFromTwoVectors; a, b need not be unit)
HD Qt so3_from_vectors(V3 a, V3 b) {
    const V3 v0 = normalized(a), v1 = normalized(b);
    double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) {
        c = c > -1.0 ? c : -1.0;
        V3 ax = cross(v0, v1);
        if (norm(ax) < 1e-300) {
            V3 e = v3(1, 0, 0);
            if (fabs(v0.x) > fabs(v0.y) && fabs(v0.x) > fabs(v0.z))
                e = v3(0, 1, 0);
            ax = cross(v0, e);
        }
        ax = normalized(ax);
        const double w2 = (1.0 + c) * 0.5;
        const double s = sqrt(1.0 - w2);
        return q_unit(Qt{sqrt(w2), ax.x * s, ax.y * s, ax.z * s});
    }
    const V3 ax = cross(v0, v1);
    const double s = sqrt((1.0 + c) * 2.0);
    const double invs = 1.0 / s;
    return q_unit(Qt{s * 0.5, ax.x * invs, ax.y * invs, ax.z * invs});
}
// V(omega): exp_SE3(omega, v) = (exp omega, V v)
HD M3 so3_V(V3 om) {
    const double th = norm(om);
    const M3 Om = skew(om);
    double A, B;
    if (th < 1e-4) {
        const double t2 = th * th;
        A = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        B = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        A = (1.0 - cos(th)) / (th * th);
        B = (th - sin(th)) / (th * th * th);
    }
    return m3_identity() + A * Om + B * (Om * Om);
}

// rotation vector of a unit quaternion, |omega| in [0, pi]
HD V3 so3_log(Qt q) {
    double qw = q.w;
    V3 v{q.x, q.y, q.z};
    if (qw < 0) {
        qw = -qw;
        v = -v;
    }
    const double n = norm(v);
    if (n < 1e-10)
        return (2.0 / qw * (1.0 - n * n / (3.0 * qw * qw))) * v;
    return (2.0 * atan2(n, qw) / n) * v;
}
HD M3 so3_Vinv(V3 om) {
    const double th = norm(om);
    const M3 Om = skew(om);
    double Cc;
    if (th < 1e-4) {
        const double t2 = th * th;
        Cc = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
    } else {
        Cc = (1.0 - 0.5 * th * sin(th) / (1.0 - cos(th))) / (th * th);
    }
    return m3_identity() - 0.5 * Om + Cc * (Om * Om);
}

// ---- SE(3)
HD Pose pose_identity() { return Pose{q_identity(), V3{0, 0, 0}}; }
HD Pose pose_mul(const Pose& a, const Pose& b) { return Pose{q_mul(a.R, b.R), a.x + q_rot(a.R, b.x)}; }
HD Pose pose_inv(const Pose& a) {
    const Qt Ri = q_inv(a.R);
    return Pose{Ri, -q_rot(Ri, a.x)};
}
HD V3 pose_act(const Pose& a, V3 p) { return q_rot(a.R, p) + a.x; }
HD Pose se3_exp(V3 om, V3 v) { return Pose{so3_exp(om), so3_V(om) * v}; }

HD void se3_log(const Pose& P, V3& om, V3& v) {
    om = so3_log(P.R);
    v = so3_Vinv(om) * P.x;
}

// 6-vector (omega, v) and 6x6 matrices (row-major) for Adjoint / adjoint algebra on the host side
struct V6 {
    V3 w, v;
};
struct M6 {
    double a[36];
};
HD V6 Ad_apply(const Pose& T, V6 U) { // Adjoint(T) U = (R w, skew(x) R w + R v)
    const V3 Rw = q_rot(T.R, U.w);
    return V6{Rw, cross(T.x, Rw) + q_rot(T.R, U.v)};
}
HD void m6_set_block(M6& Mx, int r0, int c0, const M3& B) {
    const double b[9] = {B.a00, B.a01, B.a02, B.a10, B.a11, B.a12, B.a20, B.a21, B.a22};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Mx.a[(r0 + i) * 6 + c0 + j] = b[3 * i + j];
}
HD M6 m6_zero() {
    M6 Z;
    for (int i = 0; i < 36; ++i)
        Z.a[i] = 0.0;
    return Z;
}
HD M6 se3_Adjoint(const Pose& T) { // [[R,0],[skew(x)R,R]]
    M6 A = m6_zero();
    const M3 R = q_mat(T.R);
    m6_set_block(A, 0, 0, R);
    m6_set_block(A, 3, 0, skew(T.x) * R);
    m6_set_block(A, 3, 3, R);
    return A;
}
HD M6 se3_adjoint(V6 U) { // [[skew w,0],[skew v,skew w]]
    M6 A = m6_zero();
    m6_set_block(A, 0, 0, skew(U.w));
    m6_set_block(A, 3, 0, skew(U.v));
    m6_set_block(A, 3, 3, skew(U.w));
    return A;
}
HD M6 m6_mul(const M6& A, const M6& B) {
    M6 C;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k)
                s += A.a[i * 6 + k] * B.a[k * 6 + j];
            C.a[i * 6 + j] = s;
        }
    return C;
}

// ---- pinhole camera (GIFT::PinholeCamera contract)
// ---- camera models (GIFT::PinholeCamera / StandardCamera / EquidistantCamera contracts: projectPoint, undistortPoint =
// unit bearing of a pixel, projectionJacobian = d projectPoint / d p). model: 0 pinhole, 1 radial-tangential
// (k1, k2, p1, p2, k3), 2 equidistant / Kannala-Brandt (k1..k4). See include/eqvio_types.h.
struct Cam {
    double fx, fy, cx, cy;
    int model = 0;
    double d[5] = {0, 0, 0, 0, 0};
};
// distortion of normalised image coordinates (x, y) -> (xd, yd) and its 2x2 Jacobian, row-major (j0 j1; j2 j3), returned by value.
// Round 6: named scalars in a struct returned by value, not an array - through the three camera branches the compiler kept the off-diagonal entries of `double j[4]` in SCRATCH, stored at a run-time
// selected offset (24 bytes per lane and a store -> load round trip in every kernel that projects a point: the propagation kernel's observer blocks, k_measure,
// k_outlier_stats, k_output_cov, k_stats_select, the look-ahead kernel's statistics workgroup - profiles/r05 resource usage). Same expressions, same values.
struct CamD {
    double xd, yd, j0, j1, j2, j3;
};
HD CamD cam_distort(const Cam& c, double x, double y) {
    CamD o;
    if (c.model == 1) {
        const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3], k3 = c.d[4];
        const double r2 = x * x + y * y;
        const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
        const double dr = k1 + r2 * (2.0 * k2 + r2 * 3.0 * k3); // d rad / d r2
        o.xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        o.yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        o.j0 = rad + 2.0 * x * x * dr + 2.0 * p1 * y + 6.0 * p2 * x;
        o.j1 = 2.0 * x * y * dr + 2.0 * p1 * x + 2.0 * p2 * y;
        o.j2 = o.j1;
        o.j3 = rad + 2.0 * y * y * dr + 6.0 * p1 * y + 2.0 * p2 * x;
    } else if (c.model == 2) {
        const double r = sqrt(x * x + y * y);
        const double th = atan(r), t2 = th * th;
        const double thd = th * (1.0 + t2 * (c.d[0] + t2 * (c.d[1] + t2 * (c.d[2] + t2 * c.d[3]))));
        const double dthd = 1.0 + t2 * (3.0 * c.d[0] + t2 * (5.0 * c.d[1] + t2 * (7.0 * c.d[2] + t2 * 9.0 * c.d[3])));
        if (r < 1e-8) {
            o = CamD{x, y, 1.0, 0.0, 0.0, 1.0};
        } else {
            const double sc = thd / r;
            const double dsc = (dthd / (1.0 + r * r) - sc) / r; // d sc / d r
            o.xd = sc * x;
            o.yd = sc * y;
            o.j0 = sc + dsc * x * x / r;
            o.j1 = dsc * x * y / r;
            o.j2 = o.j1;
            o.j3 = sc + dsc * y * y / r;
        }
    } else
        o = CamD{x, y, 1.0, 0.0, 0.0, 1.0};
    return o;
}
HD void cam_project(const Cam& c, V3 p, double& u, double& v) {
    const CamD o = cam_distort(c, p.x / p.z, p.y / p.z);
    u = c.fx * o.xd + c.cx;
    v = c.fy * o.yd + c.cy;
}
// pixel -> unit bearing: inverse of the distortion by Newton's method from the distorted coordinates (the map is a small
// perturbation of the identity on the image), equidistant: the scalar equation theta_d(theta) = |(xd, yd)| instead
HD V3 cam_undistort(const Cam& c, double u, double v) {
    const double xd = (u - c.cx) / c.fx, yd = (v - c.cy) / c.fy;
    if (c.model == 1) {
        double x = xd, y = yd;
        for (int it = 0; it < 12; ++it) {
            const CamD o = cam_distort(c, x, y);
            const double ex = o.xd - xd, ey = o.yd - yd;
            const double idet = 1.0 / (o.j0 * o.j3 - o.j1 * o.j2);
            x -= (o.j3 * ex - o.j1 * ey) * idet;
            y -= (o.j0 * ey - o.j2 * ex) * idet;
        }
        return normalized(V3{x, y, 1.0});
    }
    if (c.model == 2) {
        const double thd = sqrt(xd * xd + yd * yd);
        if (thd < 1e-8)
            return normalized(V3{xd, yd, 1.0});
        double th = thd;
        for (int it = 0; it < 12; ++it) {
            const double t2 = th * th;
            const double f = th * (1.0 + t2 * (c.d[0] + t2 * (c.d[1] + t2 * (c.d[2] + t2 * c.d[3])))) - thd;
            const double df = 1.0 + t2 * (3.0 * c.d[0] + t2 * (5.0 * c.d[1] + t2 * (7.0 * c.d[2] + t2 * 9.0 * c.d[3])));
            th -= f / df;
        }
        const double s = sin(th) / thd;
        return V3{s * xd, s * yd, cos(th)}; // already unit length
    }
    return normalized(V3{xd, yd, 1.0});
}
// rows of the projection Jacobian d projectPoint / d p (2x3)
HD void cam_jac(const Cam& c, V3 p, V3& j0, V3& j1) {
    const double iz = 1.0 / p.z;
    const double x = p.x * iz, y = p.y * iz;
    const CamD o = cam_distort(c, x, y);
    // d(x, y)/dp = [[iz, 0, -x iz], [0, iz, -y iz]]
    j0 = V3{c.fx * o.j0 * iz, c.fx * o.j1 * iz, -c.fx * (o.j0 * x + o.j1 * y) * iz};
    j1 = V3{c.fy * o.j2 * iz, c.fy * o.j3 * iz, -c.fy * (o.j2 * x + o.j3 * y) * iz};
}
// projection Jacobian J(p) (2x3) times skew(p): rows returned as two V3  (DRho of euclid.cpp:173-178)
HD void cam_jac_skew(const Cam& c, V3 p, V3& r0, V3& r1) {
    V3 j0, j1;
    cam_jac(c, p, j0, j1);
    // row * skew(p) = (row x ... ) : (j^T skew(p)) = -(skew(p) j)^T = -(p x j)^T = (j x p)^T
    r0 = cross(j0, p);
    r1 = cross(j1, p);
}

// ---- stereographic sphere chart differentials at the pole (src/mathematical/VIOState.cpp:246-307)
// diff0(pole)  = e3ProjectSphereDiff(R pole) * R      (2x3, returned as two rows)
// invdiff0(pole) = R^T * [[2,0],[0,2],[0,0]]          (3x2, returned as two columns)
HD void stereo_diff0(V3 pole, V3& d0, V3& d1) {
    const Qt R = so3_from_vectors(-pole, V3{0, 0, 1});
    const V3 eta = q_rot(R, pole);
    const double omz = 1.0 - eta.z;
    const double s = 1.0 / (omz * omz);
    // rows 0,1 of (I (1-eta_z) + (eta - e3) e3^T) scaled by s
    const V3 a0{omz * s, 0.0, eta.x * s};
    const V3 a1{0.0, omz * s, eta.y * s};
    const M3 Rm = q_mat(R);
    const M3 Rt = transpose(Rm);
    d0 = Rt * a0; // (a0^T R)^T
    d1 = Rt * a1;
}
HD void stereo_invdiff0(V3 pole, V3& c0, V3& c1) {
    const Qt R = so3_from_vectors(-pole, V3{0, 0, 1});
    const M3 Rt = transpose(q_mat(R));
    c0 = 2.0 * col(Rt, 0);
    c1 = 2.0 * col(Rt, 1);
}
// conv_euc2ind / conv_ind2euc (coordinateSuite/invdepth.cpp:65-81)
HD M3 conv_euc2ind(V3 q0) {
    const double rho = 1.0 / norm(q0);
    const V3 y0 = rho * q0;
    V3 d0, d1;
    stereo_diff0(y0, d0, d1);
    // rows: rho * d_k^T (I - y0 y0^T) = rho * (d_k - (d_k . y0) y0)
    const V3 r0 = rho * (d0 - dot(d0, y0) * y0);
    const V3 r1 = rho * (d1 - dot(d1, y0) * y0);
    const V3 r2 = (-rho * rho) * y0;
    return m3_rows(r0, r1, r2);
}
HD M3 conv_ind2euc(V3 q0) {
    const double rho = 1.0 / norm(q0);
    const V3 y0 = rho * q0;
    V3 c0, c1;
    stereo_invdiff0(y0, c0, c1);
    const double ir = 1.0 / rho;
    return m3_cols(ir * c0, ir * c1, (-1.0 / (rho * rho)) * y0);
}
// ind2euc in the r0 form of invdepth.cpp:201-207 / 257-262
HD M3 ind2euc_r0(V3 q0) {
    const double r0 = norm(q0);
    const V3 y0 = (1.0 / r0) * q0;
    V3 c0, c1;
    stereo_invdiff0(y0, c0, c1);
    return m3_cols(r0 * c0, r0 * c1, (-r0) * q0);
}
// sphereChart_stereo(eta, pole) (VIOState.cpp:282-287): stereographic coordinates of eta about the pole
HD void stereo_chart(V3 eta, V3 pole, double& s0, double& s1) {
    const Qt R = so3_from_vectors(-pole, V3{0, 0, 1});
    const V3 e = q_rot(R, eta);
    const double k = 1.0 / (1.0 - e.z);
    s0 = e.x * k;
    s1 = e.y * k;
}
// pointChart_invdepth / pointChart_euclid forward (VIOState.cpp:153-172): coordinates of q about q0
HD V3 point_chart(int invdepth, V3 q, V3 q0) {
    if (!invdepth)
        return q - q0;
    const double rho = 1.0 / norm(q), rho0 = 1.0 / norm(q0);
    double s0, s1;
    stereo_chart(rho * q, rho0 * q0, s0, s1);
    return V3{s0, s1, rho - rho0};
}
HD V3 e3_project_sphere_inv(double y0, double y1) { // VIOState.cpp:253-258
    const double k = 2.0 / (y0 * y0 + y1 * y1 + 1.0);
    return V3{k * y0, k * y1, 1.0 - k};
}
// pointChart_invdepth.inv (VIOState.cpp:173-186)
HD V3 invdepth_chart_inv(V3 eps, V3 q0) {
    const double rho0 = 1.0 / norm(q0);
    const V3 y0 = rho0 * q0;
    const V3 etaRot = e3_project_sphere_inv(eps.x, eps.y);
    const Qt R = so3_from_vectors(-y0, V3{0, 0, 1});
    const V3 y = q_rot(q_inv(R), etaRot);
    double rho = eps.z + rho0;
    if (rho <= 0.0)
        rho = 1e-6;
    return (1.0 / rho) * y;
}
// ---- Normal chart (sphereChart_normal / pointChart_normal, VIOState.cpp:188-209, 309-353; coordinateSuite/normal.cpp) ----------------
// The reference obtains the change of coordinates M = D(normal o euclid^-1)(0) by central differences (VIOState.cpp:391-401); it is
// block diagonal and has a closed form. With Rn = rot(y0 -> e3), r0 = |q0|, y0 = q0 / r0:
//   landmark block  M_i = [Rn.row1 / r0 ; -Rn.row0 / r0 ; -y0^T / r0],   M_i^-1 = [r0 Rn.row1^T | -r0 Rn.row0^T | -q0]
//   sensor block    identity except  [12:15, 6:9] = -skew(v0)  and  [15:21, 6:12] = Ad(T0^-1)   (inverse: +skew(v0), -Ad(T0^-1))
HD M3 normal_rot(V3 q0) { return q_mat(so3_from_vectors(normalized(q0), V3{0, 0, 1})); }
HD M3 normal_M(V3 q0) {
    const double ir = 1.0 / norm(q0);
    const M3 Rn = normal_rot(q0);
    return m3_rows(ir * row(Rn, 1), (-ir) * row(Rn, 0), (-ir * ir) * q0);
}
HD M3 normal_Minv(V3 q0) {
    const double r0 = norm(q0);
    const M3 Rn = normal_rot(q0);
    return m3_cols(r0 * row(Rn, 1), (-r0) * row(Rn, 0), -q0);
}
// sphereChart_normal.chartInvDiff0(pole) (VIOState.cpp:345-352): R^T [[0,-1],[1,0],[0,0]], as two columns
HD void normal_invdiff0(V3 q0, V3& c0, V3& c1) {
    const M3 Rn = normal_rot(q0);
    c0 = row(Rn, 1);
    c1 = -row(Rn, 0);
}
// pointChart_normal forward / inverse (VIOState.cpp:188-209)
HD V3 normal_chart(V3 q, V3 q0) {
    const double rho = 1.0 / norm(q), rho0 = 1.0 / norm(q0);
    const V3 y = normal_rot(q0) * (rho * q);
    const V3 ye3{y.y, -y.x, 0.0}; // skew(y) e3
    const double sin_th = sqrt(ye3.x * ye3.x + ye3.y * ye3.y), cos_th = y.z;
    const double th = atan2(sin_th, cos_th);
    const double k = (fabs(th) < 1e-8) ? 1.0 : th / sin_th;
    return V3{k * ye3.x, k * ye3.y, log(rho / rho0)};
}
HD V3 normal_chart_inv(V3 eps, V3 q0) {
    const double r0 = norm(q0);
    const V3 y = q_rot(so3_exp(V3{-eps.x, -eps.y, 0.0}), V3{0, 0, 1});
    return (r0 * exp(-eps.z)) * (transpose(normal_rot(q0)) * y);
}

} // namespace eqf
