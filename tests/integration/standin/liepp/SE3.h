// TEST SCAFFOLDING, not LiePP: the three group types with the members the reference's VIO types expose (R, x, a) and the
// quaternion round trip (asQuaternion / fromQuaternion / SO3d(Quaterniond)) that the reference itself uses
// (include/eqvio/csv/CSVReader.h:227-238, src/dataserver/*.cpp).
#pragma once
#include "Eigen/Dense"

namespace liepp {
class SO3d {
  public:
    SO3d() = default;
    explicit SO3d(const Eigen::Quaterniond& q) : q_(q) {}
    static SO3d Identity() { return SO3d(); }
    void setIdentity() { q_ = Eigen::Quaterniond(); }
    Eigen::Quaterniond asQuaternion() const { return q_; }
    void fromQuaternion(const Eigen::Quaterniond& q) { q_ = q; }

  private:
    Eigen::Quaterniond q_;
};
struct SE3d {
    SO3d R;
    Eigen::Vector3d x;
    SE3d() = default;
    SE3d(const SO3d& R_, const Eigen::Vector3d& x_) : R(R_), x(x_) {}
    static SE3d Identity() { return SE3d(); }
    void setIdentity() { R.setIdentity(); x.setZero(); }
};
struct SOT3d {
    SO3d R;
    double a = 1.0;
    static SOT3d Identity() { return SOT3d(); }
    void setIdentity() { R.setIdentity(); a = 1.0; }
};
} // namespace liepp
