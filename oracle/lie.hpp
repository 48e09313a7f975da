// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Lie groups used by the EqVIO hot path. The reference takes these from LiePP
// (git@github.com:pvangoor/LiePP, branch main, unpinned — .gitmodules:7-10; the submodule directory
// external/LiePP is EMPTY in /root/reference), so this file restates LiePP's published conventions,
// anchored on the reference's own call sites and pinned by the reference's property tests
// (test/test_VIOGroup.cpp, test_VIOGroupActions.cpp, test_VIOLift.cpp, test_EqFMatrices.cpp), which
// are re-stated in tests/test_oracle_properties.py:
//   SO3 : unit quaternion (w,x,y,z); R*v rotates; skew(v)w = v x w; exp/log Rodrigues.
//         SO3FromVectors(a,b) = minimal rotation with R a = b, Eigen's Quaternion::setFromTwoVectors
//         formula (call sites: src/VIOFilter.cpp:72-76, src/mathematical/VIOGroup.cpp:264-266,
//         src/mathematical/VIOState.cpp:282-307).
//   SE3 : {R,x}; (R1,x1)(R2,x2)=(R1R2, x1+R1x2); point action Rp+x; 6-vectors ordered (omega, v)
//         (src/mathematical/VIOGroup.cpp:199-201); exp(omega,v)=(exp omega, V(omega) v);
//         Adjoint=[[R,0],[skew(x)R,R]]; adjoint(U)=[[skew w,0],[skew v,skew w]].
//   SOT3: {R,a}; action a R p; product (R1R2,a1a2); inverse (R^T,1/a); exp(W)=(exp W[0:3], e^{W[3]});
//         4x4 Adjoint = blkdiag(R,1).
//   SE23: {R,x0,x1}; exp(omega,v0,v1)=(exp omega, V v0, V v1) (src/mathematical/VIOGroup.cpp:273-281).
#pragma once
#include "la.hpp"

namespace orc {

struct SO3 {
    double w = 1, x = 0, y = 0, z = 0; // unit quaternion
    static SO3 Identity() { return SO3(); }
    static SO3 fromQuat(double w, double x, double y, double z) {
        SO3 r;
        r.w = w;
        r.x = x;
        r.y = y;
        r.z = z;
        return r;
    }
    Vec3 vec() const { return vec3(x, y, z); }
    // q * v : Eigen's QuaternionBase::_transformVector
    Vec3 operator*(const Vec3& v) const {
        const Vec3 u = vec();
        Vec3 uv = cross(u, v);
        uv = uv + uv;
        return v + w * uv + cross(u, uv);
    }
    // Composition keeps the element unit-norm (re-normalised after every product). Restatement decision:
    // without it the conjugation pattern B <- T^-1 A T of liftVelocityDiscrete (VIOGroup.cpp:249) triples any
    // norm error per IMU step (|T^-1 A T| = |T|^2 when inverse() is the conjugate) and X.B blows up within a few
    // frames; LiePP's own policy cannot be checked here (source absent), the reference evidently runs for minutes.
    SO3 operator*(const SO3& o) const {
        return fromQuat(
                   w * o.w - x * o.x - y * o.y - z * o.z, w * o.x + x * o.w + y * o.z - z * o.y,
                   w * o.y + y * o.w + z * o.x - x * o.z, w * o.z + z * o.w + x * o.y - y * o.x)
            .normalizedQuat();
    }
    SO3 normalizedQuat() const {
        const double n = 1.0 / std::sqrt(w * w + x * x + y * y + z * z);
        return fromQuat(w * n, x * n, y * n, z * n);
    }
    SO3 inverse() const { return fromQuat(w, -x, -y, -z); }
    Mat3 asMatrix() const {
        Mat3 R;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R(0, 0) = 1 - (tyy + tzz);
        R(0, 1) = txy - twz;
        R(0, 2) = txz + twy;
        R(1, 0) = txy + twz;
        R(1, 1) = 1 - (txx + tzz);
        R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy;
        R(2, 1) = tyz + twx;
        R(2, 2) = 1 - (txx + tyy);
        return R;
    }
    static SO3 fromMatrix(const Mat3& m) {
        // Shepperd's method (as Eigen's quaternion-from-matrix)
        SO3 q;
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            q.w = 0.5 * t;
            t = 0.5 / t;
            q.x = (m(2, 1) - m(1, 2)) * t;
            q.y = (m(0, 2) - m(2, 0)) * t;
            q.z = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0))
                i = 1;
            if (m(2, 2) > m(i, i))
                i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            double v[3];
            v[i] = 0.5 * t;
            t = 0.5 / t;
            q.w = (m(k, j) - m(j, k)) * t;
            v[j] = (m(j, i) + m(i, j)) * t;
            v[k] = (m(k, i) + m(i, k)) * t;
            q.x = v[0];
            q.y = v[1];
            q.z = v[2];
        }
        return q;
    }
    static SO3 exp(const Vec3& omega) {
        const double th = omega.norm();
        const double half = 0.5 * th;
        double s; // sin(th/2)/th
        if (th < 1e-6) {
            const double t2 = th * th;
            s = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0;
        } else {
            s = std::sin(half) / th;
        }
        return fromQuat(std::cos(half), s * omega(0), s * omega(1), s * omega(2));
    }
    static Vec3 log(const SO3& R) {
        // angle-axis from a unit quaternion, robust for small angles; returns |omega| in [0, pi]
        double qw = R.w;
        Vec3 v = R.vec();
        if (qw < 0) {
            qw = -qw;
            v = -v;
        }
        const double n = v.norm();
        if (n < 1e-10) {
            // omega = 2*atan2(n,w)/n * v ~ 2/w * (1 - n^2/(3 w^2)) v
            return v * (2.0 / qw * (1.0 - n * n / (3.0 * qw * qw)));
        }
        const double th = 2.0 * std::atan2(n, qw);
        return v * (th / n);
    }
    // Minimal rotation R with R a = b (Eigen Quaternion::setFromTwoVectors).
    static SO3 FromVectors(const Vec3& a, const Vec3& b) {
        const Vec3 v0 = a.normalized();
        const Vec3 v1 = b.normalized();
        double c = dot(v1, v0);
        if (c < -1.0 + 1e-12) {
            // Nearly antiparallel: Eigen takes the axis from an SVD null vector of [v0^T; v1^T].
            // Restated as the unit vector orthogonal to both (sign chosen so that R v0 = v1).
            c = std::max(c, -1.0);
            Vec3 axis = cross(v0, v1);
            if (axis.norm() < 1e-300) {
                Vec3 e = vec3(1, 0, 0);
                if (std::fabs(v0(0)) > std::fabs(v0(1)) && std::fabs(v0(0)) > std::fabs(v0(2)))
                    e = vec3(0, 1, 0);
                axis = cross(v0, e);
            }
            axis = axis.normalized();
            const double w2 = (1.0 + c) * 0.5;
            const double s = std::sqrt(1.0 - w2);
            return fromQuat(std::sqrt(w2), axis(0) * s, axis(1) * s, axis(2) * s).normalizedQuat();
        }
        const Vec3 axis = cross(v0, v1);
        const double s = std::sqrt((1.0 + c) * 2.0);
        const double invs = 1.0 / s;
        return fromQuat(s * 0.5, axis(0) * invs, axis(1) * invs, axis(2) * invs).normalizedQuat();
    }
    bool hasNaN() const { return std::isnan(w) || std::isnan(x) || std::isnan(y) || std::isnan(z); }
};

// Left Jacobian-like V(omega) with exp_SE3(omega, v) = (exp omega, V v)
inline Mat3 so3_V(const Vec3& omega) {
    const double th = omega.norm();
    const Mat3 Om = skew(omega);
    double A, B; // A=(1-cos)/th^2, B=(th-sin)/th^3
    if (th < 1e-4) {
        const double t2 = th * th;
        A = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        B = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        A = (1.0 - std::cos(th)) / (th * th);
        B = (th - std::sin(th)) / (th * th * th);
    }
    return Mat3::Identity() + A * Om + B * (Om * Om);
}
inline Mat3 so3_Vinv(const Vec3& omega) {
    const double th = omega.norm();
    const Mat3 Om = skew(omega);
    double Cc; // (1/th^2)(1 - th sin / (2(1-cos)))
    if (th < 1e-4) {
        const double t2 = th * th;
        Cc = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
    } else {
        Cc = (1.0 - 0.5 * th * std::sin(th) / (1.0 - std::cos(th))) / (th * th);
    }
    return Mat3::Identity() - 0.5 * Om + Cc * (Om * Om);
}

struct SE3 {
    SO3 R;
    Vec3 x = Vec3::Zero();
    static SE3 Identity() { return SE3(); }
    SE3() = default;
    SE3(const SO3& R_, const Vec3& x_) : R(R_), x(x_) {}
    SE3 operator*(const SE3& o) const { return SE3(R * o.R, x + R * o.x); }
    Vec3 operator*(const Vec3& p) const { return R * p + x; }
    SE3 inverse() const {
        const SO3 Ri = R.inverse();
        return SE3(Ri, -(Ri * x));
    }
    Mat6 Adjoint() const {
        Mat6 Ad = Mat6::Zero();
        const Mat3 Rm = R.asMatrix();
        Ad.setBlock<3, 3>(0, 0, Rm);
        Ad.setBlock<3, 3>(3, 0, skew(x) * Rm);
        Ad.setBlock<3, 3>(3, 3, Rm);
        return Ad;
    }
    static Mat6 adjoint(const Vec6& U) {
        Mat6 ad = Mat6::Zero();
        const Mat3 Om = skew(U.block<3, 1>(0, 0));
        const Mat3 Vm = skew(U.block<3, 1>(3, 0));
        ad.setBlock<3, 3>(0, 0, Om);
        ad.setBlock<3, 3>(3, 0, Vm);
        ad.setBlock<3, 3>(3, 3, Om);
        return ad;
    }
    static SE3 exp(const Vec6& U) {
        const Vec3 om = U.block<3, 1>(0, 0);
        const Vec3 v = U.block<3, 1>(3, 0);
        return SE3(SO3::exp(om), so3_V(om) * v);
    }
    static Vec6 log(const SE3& P) {
        const Vec3 om = SO3::log(P.R);
        const Vec3 v = so3_Vinv(om) * P.x;
        Vec6 U;
        U.setBlock<3, 1>(0, 0, om);
        U.setBlock<3, 1>(3, 0, v);
        return U;
    }
    bool hasNaN() const { return R.hasNaN() || x.hasNaN(); }
};

struct SOT3 {
    SO3 R;
    double a = 1.0;
    static SOT3 Identity() { return SOT3(); }
    SOT3() = default;
    SOT3(const SO3& R_, double a_) : R(R_), a(a_) {}
    SOT3 operator*(const SOT3& o) const { return SOT3(R * o.R, a * o.a); }
    Vec3 operator*(const Vec3& p) const { return a * (R * p); }
    SOT3 inverse() const { return SOT3(R.inverse(), 1.0 / a); }
    static SOT3 exp(const Vec4& W) { return SOT3(SO3::exp(W.block<3, 1>(0, 0)), std::exp(W(3))); }
    static Vec4 log(const SOT3& Q) {
        Vec4 W;
        W.setBlock<3, 1>(0, 0, SO3::log(Q.R));
        W(3) = std::log(Q.a);
        return W;
    }
    M<4, 4> Adjoint() const {
        M<4, 4> Ad = M<4, 4>::Zero();
        Ad.setBlock<3, 3>(0, 0, R.asMatrix());
        Ad(3, 3) = 1.0;
        return Ad;
    }
    bool hasNaN() const { return R.hasNaN() || std::isnan(a); }
};

struct SE23 {
    SO3 R;
    Vec3 x0 = Vec3::Zero(), x1 = Vec3::Zero();
    static SE23 exp(const M<9, 1>& U) {
        const Vec3 om = U.block<3, 1>(0, 0);
        const Mat3 V = so3_V(om);
        SE23 r;
        r.R = SO3::exp(om);
        r.x0 = V * U.block<3, 1>(3, 0);
        r.x1 = V * U.block<3, 1>(6, 0);
        return r;
    }
    static M<9, 1> log(const SE23& P) {
        const Vec3 om = SO3::log(P.R);
        const Mat3 Vi = so3_Vinv(om);
        M<9, 1> U;
        U.setBlock<3, 1>(0, 0, om);
        U.setBlock<3, 1>(3, 0, Vi * P.x0);
        U.setBlock<3, 1>(6, 0, Vi * P.x1);
        return U;
    }
};

} // namespace orc
