// TEST SCAFFOLDING, not GIFT: camera classes carrying what the binding needs from a GIFT camera (image size, fx, fy, cx, cy,
// distortion coefficients) and the class names the reference instantiates (PinholeCamera: SimulationDataServer.cpp:175,
// StandardCamera = radtan: ASLDatasetReader.cpp:93, EquidistantCamera: UZHFPVDatasetReader.cpp:102).
#pragma once
#include "Eigen/Dense"
#include <array>
#include <memory>
#include <vector>

namespace GIFT {
struct ImageSize {
    int width = 0, height = 0;
};
class GICamera {
  public:
    virtual ~GICamera() = default;
    ImageSize imageSize;
    double fx = 1, fy = 1, cx = 0, cy = 0;
    // pinhole forms (what VIOFilter::removeOutliers / addNewLandmarks call on the simulator's camera); distorted models: not needed by the driver
    virtual Eigen::Vector2d projectPoint(const Eigen::Vector3d& p) const { return Eigen::Vector2d(fx * p.x() / p.z() + cx, fy * p.y() / p.z() + cy); }
    virtual Eigen::Vector3d undistortPoint(const Eigen::Vector2d& y) const {
        const Eigen::Vector3d b((y.x() - cx) / fx, (y.y() - cy) / fy, 1.0);
        return b * (1.0 / b.norm());
    }
};
class PinholeCamera : public GICamera {
  public:
    PinholeCamera(ImageSize sz, double fx_, double fy_, double cx_, double cy_) { imageSize = sz; fx = fx_; fy = fy_; cx = cx_; cy = cy_; }
};
class StandardCamera : public GICamera { // radial-tangential (k1, k2, p1, p2[, k3])
  public:
    std::vector<double> dist;
};
class EquidistantCamera : public GICamera { // Kannala-Brandt (k1..k4)
  public:
    std::array<double, 4> dist{};
};
using GICameraPtr = std::shared_ptr<const GICamera>;
} // namespace GIFT
