// See VIOWriter.hpp. Formats restated from src/VIOWriter.cpp (line numbers cited per function).
#include "VIOWriter.hpp"
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <sys/stat.h>

namespace eqvio_amd {
using namespace eqf;

namespace {
// One CSV line: elements formatted one by one with a default-precision stream, joined by ", " (CSVLine.h:84-89, 153-161)
class Line {
    std::vector<std::string> data;

  public:
    template <typename T> Line& operator<<(const T& d) {
        std::stringstream ss;
        ss << d;
        data.emplace_back(ss.str());
        return *this;
    }
    Line& operator<<(const V3& v) { return *this << v.x << v.y << v.z; }
    Line& operator<<(const Qt& q) { return *this << q.w << q.x << q.y << q.z; }
    Line& operator<<(const Pose& P) { return *this << P.x << P.R; } // CSVLine.h:213
    friend std::ostream& operator<<(std::ostream& os, const Line& l) {
        for (size_t i = 0; i < l.data.size(); ++i)
            os << (i ? ", " : "") << l.data[i];
        return os;
    }
};
void stampPrefix(std::ofstream& f, double stamp) { f << std::setprecision(20) << stamp << ", " << std::setprecision(6); }
void makeDirs(const std::string& path) { // std::filesystem::create_directories
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/')
            mkdir(path.substr(0, i).c_str(), 0755);
}
// x^T M^-1 x for a small symmetric positive definite M (k <= 6), Gauss elimination with partial pivoting
template <int K> double quadInv(const double (&M)[K][K], const double (&x)[K]) {
    double a[K][K + 1];
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < K; ++j)
            a[i][j] = M[i][j];
        a[i][K] = x[i];
    }
    for (int c = 0; c < K; ++c) {
        int p = c;
        for (int r = c + 1; r < K; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[p][c]))
                p = r;
        for (int j = 0; j <= K; ++j)
            std::swap(a[p][j], a[c][j]);
        for (int r = c + 1; r < K; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int j = c; j <= K; ++j)
                a[r][j] -= f * a[c][j];
        }
    }
    double y[K];
    for (int i = K - 1; i >= 0; --i) {
        double s = a[i][K];
        for (int j = i + 1; j < K; ++j)
            s -= a[i][j] * y[j];
        y[i] = s / a[i][i];
    }
    double q = 0;
    for (int i = 0; i < K; ++i)
        q += x[i] * y[i];
    return q;
}
} // namespace

VIOWriter::VIOWriter(const std::string& providedOutputDir) : outputDir(providedOutputDir) { // :22-31
    if (outputDir.empty() || outputDir.back() != '/')
        outputDir += "/";
    makeDirs(outputDir);
}

void VIOWriter::writeStates(const double& stamp, const VIOState& xi) { // :33-80
    if (!IMUStateFile.is_open()) {
        IMUStateFile.open(outputDir + "IMUState.csv");
        IMUStateFile << "time, px, py, pz, qw, qx, qy, qz, vx, vy, vz\n";
    }
    stampPrefix(IMUStateFile, stamp);
    IMUStateFile << (Line() << xi.sensor.pose << xi.sensor.velocity) << '\n';

    if (!cameraFile.is_open()) {
        cameraFile.open(outputDir + "camera.csv");
        cameraFile << "time, px, py, pz, qw, qx, qy, qz\n";
    }
    stampPrefix(cameraFile, stamp);
    cameraFile << (Line() << xi.sensor.cameraOffset) << '\n';

    if (!biasFile.is_open()) {
        biasFile.open(outputDir + "bias.csv");
        biasFile << "time, bias_gyr_x, bias_gyr_y, bias_gyr_z, bias_acc_x, bias_acc_y, bias_acc_z\n";
    }
    stampPrefix(biasFile, stamp);
    {
        Line line;
        for (double b : xi.sensor.inputBias)
            line << b;
        biasFile << line << '\n';
    }

    if (!pointsFile.is_open()) {
        pointsFile.open(outputDir + "points.csv");
        pointsFile << "time, p1id, p1x, p1y, p1z, ...\n";
    }
    stampPrefix(pointsFile, stamp);
    {
        Line line;
        const Pose PC = pose_mul(xi.sensor.pose, xi.sensor.cameraOffset); // world-frame points
        for (const Landmark& q : xi.cameraLandmarks)
            line << q.id << pose_act(PC, q.p);
        pointsFile << line << '\n';
    }
}

void VIOWriter::writeFeatures(const VisionMeasurement& y) { // :82-94
    if (!featuresFile.is_open()) {
        featuresFile.open(outputDir + "features.csv");
        featuresFile << "time, z1id, z1x, z1y, ...\n";
    }
    stampPrefix(featuresFile, y.stamp);
    Line line;
    for (const auto& kv : y.camCoordinates)
        line << kv.first << kv.second[0] << kv.second[1];
    featuresFile << line << '\n';
}

void VIOWriter::writeTiming(const LoopTimer::LoopTimingData& timingData) { // :96-115, header = labels in map order
    if (!timingFile.is_open()) {
        timingFile.open(outputDir + "timing.csv");
        Line header;
        header << "time";
        for (const auto& kv : timingData.timings)
            header << kv.first;
        timingFile << header << '\n';
    }
    stampPrefix(timingFile, timingData.loopTimeStart.count());
    Line line;
    for (const auto& kv : timingData.timings)
        line << kv.second.count();
    timingFile << line << '\n';
}

void VIOWriter::writeLandmarkError(const double& stamp, const VIOState& trueState, const VIOState& estState) { // :117-138
    if (!landmarkErrorFile.is_open()) {
        landmarkErrorFile.open(outputDir + "landmarkError.csv");
        landmarkErrorFile << "time, lm_err_1, lm_err_2, ...\n";
    }
    stampPrefix(landmarkErrorFile, stamp);
    // one column per TRUE landmark, NaN where the filter does not hold it
    std::map<int, V3> est;
    for (const Landmark& lm : estState.cameraLandmarks)
        est.emplace(lm.id, lm.p);
    Line line;
    for (const Landmark& lm : trueState.cameraLandmarks) {
        const auto it = est.find(lm.id);
        line << (it == est.end() ? std::nan("") : norm(it->second - lm.p));
    }
    landmarkErrorFile << line << '\n';
}

void VIOWriter::writeConsistency(const double& stamp, const VIOState& trueState, const VIO_eqf& filter) { // :140-228
    if (!trueStateFile.is_open()) {
        trueStateFile.open(outputDir + "trueState.csv");
        // (the header text, missing comma after bias_acc_z included, is the reference's)
        trueStateFile << "time, pose_tx, pose_ty, pose_tz, pose_qw, pose_qx, pose_qy, pose_qz,"
                         "pose_vx, pose_vy, pose_vz, cam_tx, cam_ty, cam_tz, cam_qw, cam_qx, cam_qy, cam_qz,"
                         "bias_gyr_x, bias_gyr_y, bias_gyr_z, bias_acc_x, bias_acc_y, bias_acc_z"
                         "num_lm, lm_1_id, lm_1_x, lm_1_y, lm_1_z, lm_2_id, lm_2_x, lm_2_y, lm_2_z, ...\n";
    }
    stampPrefix(trueStateFile, stamp);
    {
        Line line; // VIOState.cpp:80-92: sensor (pose, velocity, cameraOffset, bias), count, then id + point
        line << trueState.sensor.pose << trueState.sensor.velocity << trueState.sensor.cameraOffset;
        for (double b : trueState.sensor.inputBias)
            line << b;
        line << trueState.cameraLandmarks.size();
        for (const Landmark& lm : trueState.cameraLandmarks)
            line << lm.id << lm.p;
        trueStateFile << line << '\n';
    }

    const VIOState xi0 = filter.xi0();
    const VIOGroup X = filter.X();
    double S[21][21]; // the sensor block of Sigma
    {
        std::vector<double> blk(21 * 21);
        if (eqf_get_sigma_block(filter.ctx, 0, 0, 21, 21, blk.data()) != 0)
            throw std::runtime_error("VIOWriter: eqf_get_sigma_block failed");
        for (int i = 0; i < 21; ++i)
            for (int j = 0; j < 21; ++j)
                S[i][j] = blk[(size_t)j * 21 + i];
    }
    const Pose errorPose = pose_mul(trueState.sensor.pose, pose_inv(X.A));
    V3 epsR, epsX;
    se3_log(pose_mul(pose_inv(xi0.sensor.pose), errorPose), epsR, epsX);

    if (!neesFile.is_open()) {
        neesFile.open(outputDir + "nees.csv");
        neesFile << "time, NEES, DoF, PoseNEES, AttitudeNEES\n";
    }
    stampPrefix(neesFile, stamp);
    {
        double fullNEES = std::nan("");
        try {
            fullNEES = filter.computeNEES(trueState);
        } catch (const std::exception&) {
            // Sigma positive definite only up to rounding: the device factorisation refuses it (see main_sim.cpp); NaN in the file
        }
        double P6[6][6], e6[6] = {epsR.x, epsR.y, epsR.z, epsX.x, epsX.y, epsX.z};
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
                P6[i][j] = S[6 + i][6 + j];
        const double poseNEES = quadInv<6>(P6, e6);
        const V3 attEps = so3_log(q_mul(q_inv(xi0.sensor.pose.R), errorPose.R));
        double P3[3][3], e3[3] = {attEps.x, attEps.y, attEps.z};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                P3[i][j] = S[6 + i][6 + j];
        const double attitudeNEES = quadInv<3>(P3, e3);
        neesFile << (Line() << fullNEES << xi0.Dim() << poseNEES << attitudeNEES) << '\n';
    }

    auto consistency = [&](std::ofstream& f, const char* name, const char* header, const double (&eps)[6], int s0) {
        if (!f.is_open()) {
            f.open(outputDir + name);
            f << header;
        }
        stampPrefix(f, stamp);
        Line line;
        for (double e : eps)
            line << e;
        for (int k = 0; k < 6; ++k)
            line << S[s0 + k][s0 + k];
        f << line << '\n';
    };
    {
        const double eps[6] = {epsR.x, epsR.y, epsR.z, epsX.x, epsX.y, epsX.z};
        consistency(poseConsistencyFile, "poseConsistency.csv",
                    "time, eps_rx, eps_ry, eps_rz, eps_px, eps_py, eps_pz,Sigma2_rx, Sigma2_ry, Sigma2_rz, Sigma2_px, Sigma2_py, Sigma2_pz\n", eps, 6);
    }
    {
        const Pose errorCamera = pose_mul(pose_mul(X.A, trueState.sensor.cameraOffset), pose_inv(X.B));
        V3 r, x;
        se3_log(pose_mul(pose_inv(xi0.sensor.cameraOffset), errorCamera), r, x);
        const double eps[6] = {r.x, r.y, r.z, x.x, x.y, x.z};
        consistency(cameraConsistencyFile, "cameraConsistency.csv",
                    "time, eps_rx, eps_ry, eps_rz, eps_px, eps_py, eps_pz,Sigma2_rx, Sigma2_ry, Sigma2_rz, Sigma2_px, Sigma2_py, Sigma2_pz\n", eps, 15);
    }
    {
        double eps[6];
        for (int k = 0; k < 6; ++k)
            eps[k] = trueState.sensor.inputBias[k] - X.beta[k] - xi0.sensor.inputBias[k];
        consistency(biasConsistencyFile, "biasConsistency.csv",
                    "time, eps_gyr_x, eps_gyr_y, eps_gyr_z, eps_acc_x, eps_acc_y, eps_acc_z,Sigma2_gyr_x, Sigma2_gyr_y, Sigma2_gyr_z, Sigma2_acc_x, Sigma2_acc_y, "
                    "Sigma2_acc_z\n",
                    eps, 0);
    }
}

} // namespace eqvio_amd
