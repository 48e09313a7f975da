// Driver for tests/test_integration_stub.py: reference-style caller code (aggregate initialisation, copies, direct readers of
// Sigma) over the binding VIO_eqf_mi355x.cpp. Reads "name count v0 v1 ..." records, writes the same.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "eqvio/mathematical/VIO_eqf.h"

const EqFCoordinateSuite EqFCoordinateSuite_euclid{}, EqFCoordinateSuite_invdepth{}, EqFCoordinateSuite_normal{}; // the reference defines these in coordinateSuite/*.cpp

using Rec = std::map<std::string, std::vector<double>>;
static Rec readRecords(const char* path) {
    Rec r;
    std::ifstream in(path);
    std::string name;
    size_t cnt;
    while (in >> name >> cnt) {
        std::vector<double>& v = r[name];
        v.resize(cnt);
        for (double& x : v)
            in >> x;
    }
    return r;
}
static std::ofstream out;
static void put(const std::string& name, const double* v, size_t cnt) {
    out << name << ' ' << cnt;
    char buf[40];
    for (size_t i = 0; i < cnt; ++i) {
        std::snprintf(buf, sizeof buf, " %.17g", v[i]);
        out << buf;
    }
    out << '\n';
}
static liepp::SE3d pose7(const double* q) { return liepp::SE3d(liepp::SO3d(Eigen::Quaterniond(q[0], q[1], q[2], q[3])), Eigen::Vector3d(q[4], q[5], q[6])); }
static VIOSensorState sensor23(const double* v) {
    VIOSensorState s;
    for (int i = 0; i < 6; ++i) s.inputBias(i) = v[i];
    s.pose = pose7(v + 6);
    s.velocity = Eigen::Vector3d(v[13], v[14], v[15]);
    s.cameraOffset = pose7(v + 16);
    return s;
}
static void putSensor(const std::string& name, const VIOSensorState& s) {
    double v[23];
    for (int i = 0; i < 6; ++i) v[i] = s.inputBias(i);
    auto pose = [](const liepp::SE3d& P, double* q) {
        const Eigen::Quaterniond a = P.R.asQuaternion();
        q[0] = a.w(), q[1] = a.x(), q[2] = a.y(), q[3] = a.z(), q[4] = P.x(0), q[5] = P.x(1), q[6] = P.x(2);
    };
    pose(s.pose, v + 6);
    for (int i = 0; i < 3; ++i) v[13 + i] = s.velocity(i);
    pose(s.cameraOffset, v + 16);
    put(name, v, 23);
}
static void putState(const std::string& tag, const VIOState& xi) {
    putSensor(tag + "_sensor", xi.sensor);
    std::vector<double> ids, p;
    for (const Landmark& lm : xi.cameraLandmarks) {
        ids.push_back(lm.id);
        p.insert(p.end(), lm.p.data(), lm.p.data() + 3);
    }
    put(tag + "_ids", ids.data(), ids.size());
    put(tag + "_p", p.data(), p.size());
}
// what the writers read from a const VIO_eqf& (VIOWriter.cpp:171-222)
static void putReaders(const std::string& tag, const VIO_eqf& filter) {
    const Eigen::Matrix<double, 6, 6> poseCov = filter.Sigma.block<6, 6>(6, 6);
    put(tag + "_poseCov", poseCov.data(), 36);
    const Eigen::Matrix<double, 3, 3> attCov = filter.Sigma.block<3, 3>(6, 6);
    put(tag + "_attCov", attCov.data(), 9);
    const Eigen::Vector<double, 6> sigmaPose = filter.Sigma.diagonal().segment<6>(6), sigmaCamera = filter.Sigma.diagonal().segment<6>(15),
                                   sigmaBias = filter.Sigma.diagonal().segment<6>(0);
    put(tag + "_sigmaPose", sigmaPose.data(), 6);
    put(tag + "_sigmaCamera", sigmaCamera.data(), 6);
    put(tag + "_sigmaBias", sigmaBias.data(), 6);
    put(tag + "_Sigma", filter.Sigma.data(), (size_t)filter.Sigma.rows() * filter.Sigma.cols());
    std::vector<double> ids(filter.X.id.begin(), filter.X.id.end());
    put(tag + "_Xid", ids.data(), ids.size());
}

int main(int argc, char** argv) {
    if (argc != 3) {
        std::cerr << "usage: run_one_frame <in> <out>\n";
        return 2;
    }
    Rec in = readRecords(argv[1]);
    out.open(argv[2]);
    const int chart = (int)in["chart"][0], N = (int)in["ids"].size(), k = (int)in["dts"].size();
    const EqFCoordinateSuite* suite = chart == 0 ? &EqFCoordinateSuite_euclid : chart == 1 ? &EqFCoordinateSuite_invdepth : &EqFCoordinateSuite_normal;

    // ---- test/test_FilterStatistics.cpp:40: aggregate initialisation
    VIOState xi0;
    xi0.sensor = sensor23(in["xi0"].data());
    std::vector<int> ids;
    for (int i = 0; i < N; ++i) {
        Landmark lm;
        lm.id = (int)in["ids"][i];
        lm.p = Eigen::Vector3d(in["p"][3 * i], in["p"][3 * i + 1], in["p"][3 * i + 2]);
        xi0.cameraLandmarks.push_back(lm);
        ids.push_back(lm.id);
    }
    const int n = 21 + 3 * N;
    Eigen::MatrixXd Sigma0(n, n);
    std::copy(in["Sigma0"].begin(), in["Sigma0"].end(), Sigma0.data());
    VIOGroup X0 = VIOGroup::Identity(ids);
    {
        const double* g = in["Xs"].data();
        for (int i = 0; i < 6; ++i) X0.beta(i) = g[i];
        X0.A = pose7(g + 6), X0.w = Eigen::Vector3d(g[13], g[14], g[15]), X0.B = pose7(g + 16);
        for (int i = 0; i < N; ++i) {
            const double* q = &in["Q"][5 * i];
            X0.Q[i].R = liepp::SO3d(Eigen::Quaterniond(q[0], q[1], q[2], q[3]));
            X0.Q[i].a = q[4];
        }
    }
    VIO_eqf filter;
    filter = VIO_eqf{suite, xi0, X0, Sigma0};

    Eigen::Matrix<double, 12, 12> Q;
    for (int i = 0; i < 12; ++i) Q(i, i) = in["Qdiag12"][i];
    Eigen::MatrixXd P = Eigen::MatrixXd::Zero(n, n);
    for (int i = 0; i < n; ++i) P(i, i) = i < 21 ? in["Pdiag8"][i / 3] : in["Pdiag8"][7];

    // ---- VIOFilter::integrateUpToTime, per sample (VIOFilter.cpp:150-178)
    for (int s = 0; s < k; ++s) {
        const double* v = &in["imus"][13 * s];
        IMUVelocity imu;
        imu.stamp = v[0];
        imu.gyr = Eigen::Vector3d(v[1], v[2], v[3]), imu.acc = Eigen::Vector3d(v[4], v[5], v[6]);
        imu.gyrBiasVel = Eigen::Vector3d(v[7], v[8], v[9]), imu.accBiasVel = Eigen::Vector3d(v[10], v[11], v[12]);
        filter.integrateRiccatiStateFast(imu, in["dts"][s], Q, P);
        filter.integrateObserverState(imu, in["dts"][s], true);
    }
    VIO_eqf fork = filter; // copy while the device is ahead of the host members

    // ---- VIOFilter::processVisionData: the update (VIOFilter.cpp:229-233)
    VisionMeasurement y;
    const std::vector<double>& c = in["cam"];
    y.cameraPtr = std::make_shared<GIFT::PinholeCamera>(GIFT::ImageSize{(int)c[4], (int)c[5]}, c[0], c[1], c[2], c[3]);
    for (size_t i = 0; i < in["meas_ids"].size(); ++i)
        y.camCoordinates[(int)in["meas_ids"][i]] = Eigen::Vector2d(in["meas_y"][2 * i], in["meas_y"][2 * i + 1]);
    const int m = 2 * (int)y.camCoordinates.size();
    const Eigen::MatrixXd R = Eigen::MatrixXd::Identity(m, m) * in["meas_var"][0];
    const int probe = y.camCoordinates.begin()->first;
    const Eigen::Matrix2d outCov = filter.getOutputCovById(probe, y.camCoordinates.begin()->second, y.cameraPtr);
    put("a_outputCov", outCov.data(), 4);
    filter.performVisionUpdate(y, R, true, false);

    // ---- readers: stateEstimate, NEES, and the writers' direct member access behind viewEqFState()
    putState("a_est", filter.stateEstimate());
    VIOState truth;
    truth.sensor = sensor23(in["truth_sensor"].data());
    for (int i = 0; i < N; ++i) {
        Landmark lm;
        lm.id = ids[i];
        lm.p = Eigen::Vector3d(in["truth_p"][3 * i], in["truth_p"][3 * i + 1], in["truth_p"][3 * i + 2]);
        truth.cameraLandmarks.push_back(lm);
    }
    const double nees = filter.computeNEES(truth);
    put("a_nees", &nees, 1);
    const Eigen::Matrix3d lmCov = filter.getLandmarkCovById(probe);
    put("a_landmarkCov", lmCov.data(), 9);
    filter.pull(); // = the first line of VIOFilter::viewEqFState() in the bound tree
    const VIO_eqf& view = filter;
    putReaders("a", view);

    // ---- the fork continues on its own: bookkeeping + update with the landmarks it has left
    fork.removeLandmarkById(ids[1]);
    std::vector<Landmark> fresh(1);
    fresh[0].id = 100000;
    fresh[0].p = Eigen::Vector3d(in["new_p"][0], in["new_p"][1], in["new_p"][2]);
    fork.addNewLandmarks(fresh, Eigen::MatrixXd::Identity(3, 3) * in["new_var"][0]);
    VisionMeasurement y2 = y;
    y2.camCoordinates.erase(ids[1]);
    const int m2 = 2 * (int)y2.camCoordinates.size();
    fork.performVisionUpdate(y2, Eigen::MatrixXd::Identity(m2, m2) * in["meas_var"][0], true, false);
    putState("b_est", fork.stateEstimate());
    fork.pull();
    putReaders("b", fork);

    // ---- a copy of a filter whose host members are current, edited on the host, used again
    VIO_eqf third = filter;
    third.Sigma(0, 0) *= 2.0; // direct assignment, as VIOFilter::setState does (VIOFilter.cpp:81-98)
    third.markHostEdited();
    third.integrateObserverState(IMUVelocity{}, 0.0, true); // dt = 0: uploads, changes nothing
    const Eigen::Matrix3d lm3 = third.getLandmarkCovById(probe);
    put("c_landmarkCov", lm3.data(), 9);
    third.removeLandmarkByIndex(0);
    third.pull();
    putReaders("c", third);
    return 0;
}
