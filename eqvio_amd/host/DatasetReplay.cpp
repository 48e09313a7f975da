// See DatasetReplay.hpp.
#include "DatasetReplay.hpp"
#include <cctype>
#include <cmath>
#include <sstream>

namespace eqvio_amd {

std::vector<std::string> splitLine(const std::string& line, char delim) {
    std::vector<std::string> out;
    std::string cell;
    std::stringstream ss(line);
    while (std::getline(ss, cell, delim)) {
        size_t a = 0, b = cell.size();
        while (a < b && (cell[a] == ' ' || cell[a] == '\t'))
            ++a;
        while (b > a && (cell[b - 1] == ' ' || cell[b - 1] == '\t' || cell[b - 1] == '\r'))
            --b;
        if (delim == ' ' && a == b)
            continue; // runs of blanks
        out.emplace_back(cell.substr(a, b - a));
    }
    return out;
}

namespace {
char delimOf(DatasetFormat f) { return f == DatasetFormat::ASL ? ',' : ' '; }
bool nextDataLine(std::ifstream& f, std::string& line) {
    while (std::getline(f, line))
        if (!line.empty() && line != "\r")
            return true;
    return false;
}
} // namespace

TrackReplayServer::TrackReplayServer(const std::string& imuFileName, const std::string& featuresFileName, DatasetFormat fmt, const GICameraPtr& cam, double lag)
    : imuFile(imuFileName), featuresFile(featuresFileName), format(fmt), cameraPtr(cam), cameraLag(lag) {
    if (!imuFile)
        throw std::runtime_error("TrackReplayServer: cannot open " + imuFileName);
    if (!featuresFile)
        throw std::runtime_error("TrackReplayServer: cannot open " + featuresFileName);
    std::string header;
    std::getline(imuFile, header);      // skip the header (ASLDatasetReader.cpp:28, UZHFPVDatasetReader.cpp:28)
    std::getline(featuresFile, header); // "time, z1id, z1x, z1y, ..."
    nextImageData = readFeatures();
    nextIMUData = readIMU();
}

std::unique_ptr<IMUVelocity> TrackReplayServer::readIMU() {
    std::string line;
    if (!nextDataLine(imuFile, line))
        return nullptr;
    const std::vector<std::string> c = splitLine(line, delimOf(format));
    const size_t o = format == DatasetFormat::UZHFPV ? 1 : 0; // leading index column
    if (c.size() < o + 7)
        throw std::runtime_error("TrackReplayServer: short IMU line: " + line);
    auto num = [&c, o](size_t k) { return std::stod(c[o + k]); };
    IMUVelocity v;
    v.stamp = num(0) * (format == DatasetFormat::ASL ? 1e-9 : 1.0);
    v.gyr = V3{num(1), num(2), num(3)};
    v.acc = V3{num(4), num(5), num(6)};
    if (c.size() >= o + 13) {
        v.gyrBiasVel = V3{num(7), num(8), num(9)};
        v.accBiasVel = V3{num(10), num(11), num(12)};
    }
    return std::make_unique<IMUVelocity>(v);
}

std::unique_ptr<VisionMeasurement> TrackReplayServer::readFeatures() {
    std::string line;
    if (!nextDataLine(featuresFile, line))
        return nullptr;
    std::vector<std::string> c = splitLine(line, ',');
    while (c.size() > 1 && c.back().empty()) // a frame without features is written as "stamp, " (VIOWriter.cpp:88-93)
        c.pop_back();
    if (c.empty() || (c.size() - 1) % 3 != 0)
        throw std::runtime_error("TrackReplayServer: malformed feature line: " + line);
    auto m = std::make_unique<VisionMeasurement>();
    m->stamp = std::stod(c[0]) - cameraLag;
    m->cameraPtr = cameraPtr;
    for (size_t k = 1; k + 2 <= c.size() - 1; k += 3)
        m->camCoordinates[std::stoi(c[k])] = {std::stod(c[k + 1]), std::stod(c[k + 2])};
    return m;
}

MeasurementType TrackReplayServer::nextMeasurementType() const {
    if (nextImageData && nextIMUData)
        return nextImageData->stamp <= nextIMUData->stamp ? MeasurementType::Image : MeasurementType::IMU;
    if (nextImageData)
        return MeasurementType::Image;
    if (nextIMUData)
        return MeasurementType::IMU;
    return MeasurementType::None;
}
double TrackReplayServer::nextTime() const {
    switch (nextMeasurementType()) {
    case MeasurementType::Image: return nextImageData->stamp;
    case MeasurementType::IMU: return nextIMUData->stamp;
    default: return std::nan("");
    }
}
IMUVelocity TrackReplayServer::getIMU() {
    const IMUVelocity r = *nextIMUData;
    nextIMUData = readIMU();
    return r;
}
VisionMeasurement TrackReplayServer::getSimVision() {
    const VisionMeasurement r = *nextImageData;
    nextImageData = readFeatures();
    return r;
}

std::vector<StampedPose> TrackReplayServer::groundtruth(const std::string& fileName, DatasetFormat format) {
    std::ifstream f(fileName);
    if (!f)
        throw std::runtime_error("TrackReplayServer: cannot open " + fileName);
    std::string line;
    std::getline(f, line); // header
    std::vector<StampedPose> poses;
    double prevPoseTime = -1e8;
    while (nextDataLine(f, line)) {
        const std::vector<std::string> c = splitLine(line, delimOf(format));
        if (c.size() < 8)
            throw std::runtime_error("TrackReplayServer: short ground-truth line: " + line);
        StampedPose p;
        p.t = std::stod(c[0]) * (format == DatasetFormat::ASL ? 1e-9 : 1.0);
        p.pose.x = V3{std::stod(c[1]), std::stod(c[2]), std::stod(c[3])};
        p.pose.R = eqf::q_unit(Qt{std::stod(c[4]), std::stod(c[5]), std::stod(c[6]), std::stod(c[7])});
        if (p.t > prevPoseTime + 1e-8) { // avoid poses with the same timestamp
            poses.emplace_back(p);
            prevPoseTime = p.t;
        }
    }
    return poses;
}

namespace {
// the numbers that follow `key:` in the text (brackets, commas, dashes and line breaks in between are skipped), at most `want`; stops at the next key
std::vector<double> numbersAfterKey(const std::string& text, size_t from, const std::string& key, size_t want, const std::string& fileName) {
    size_t at = text.find(key + ":", from);
    if (at == std::string::npos)
        throw std::runtime_error("readCameraFile: no key '" + key + "' in " + fileName);
    at += key.size() + 1;
    std::vector<double> out;
    while (at < text.size() && out.size() < want) {
        const char ch = text[at];
        if (std::isdigit((unsigned char)ch) || ((ch == '-' || ch == '+' || ch == '.') && at + 1 < text.size() && (std::isdigit((unsigned char)text[at + 1]) || text[at + 1] == '.'))) {
            size_t used = 0;
            out.push_back(std::stod(text.substr(at), &used));
            at += used;
        } else if (std::isalpha((unsigned char)ch) || ch == '_') {
            break; // the next key
        } else
            ++at;
    }
    if (out.size() < want)
        throw std::runtime_error("readCameraFile: key '" + key + "' of " + fileName + " has " + std::to_string(out.size()) + " numbers, expected " + std::to_string(want));
    return out;
}
Pose poseFromRowMajor(const std::vector<double>& T) { // homogeneous 4 x 4 -> (unit quaternion, translation); the rotation block by Shepperd's method
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
    const double tr = m00 + m11 + m22;
    Qt q;
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q = Qt{0.25 * s, (m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s};
    } else if (m00 > m11 && m00 > m22) {
        const double s = std::sqrt(1.0 + m00 - m11 - m22) * 2;
        q = Qt{(m21 - m12) / s, 0.25 * s, (m01 + m10) / s, (m02 + m20) / s};
    } else if (m11 > m22) {
        const double s = std::sqrt(1.0 + m11 - m00 - m22) * 2;
        q = Qt{(m02 - m20) / s, (m01 + m10) / s, 0.25 * s, (m12 + m21) / s};
    } else {
        const double s = std::sqrt(1.0 + m22 - m00 - m11) * 2;
        q = Qt{(m10 - m01) / s, (m02 + m20) / s, (m12 + m21) / s, 0.25 * s};
    }
    return Pose{eqf::q_unit(q), V3{T[3], T[7], T[11]}};
}
} // namespace

void readCameraFile(const std::string& fileName, DatasetFormat format, Camera& camera, Pose& cameraOffset) {
    std::ifstream f(fileName);
    if (!f)
        throw std::runtime_error("readCameraFile: cannot open " + fileName);
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string text = buf.str();
    size_t from = 0;
    if (format == DatasetFormat::UZHFPV) {
        from = text.find("cam0:");
        if (from == std::string::npos)
            throw std::runtime_error("readCameraFile: no 'cam0' node in " + fileName);
    }
    const std::vector<double> res = numbersAfterKey(text, from, "resolution", 2, fileName), K = numbersAfterKey(text, from, "intrinsics", 4, fileName);
    camera.c = eqvio_camera{};
    camera.c.width = (int)res[0], camera.c.height = (int)res[1];
    camera.c.fx = K[0], camera.c.fy = K[1], camera.c.cx = K[2], camera.c.cy = K[3];
    if (format == DatasetFormat::ASL) {
        camera.c.model = EQVIO_CAMERA_RADTAN;
        std::vector<double> d;
        try {
            d = numbersAfterKey(text, from, "distortion_coefficients", 5, fileName);
        } catch (const std::runtime_error&) {
            d = numbersAfterKey(text, from, "distortion_coefficients", 4, fileName); // EuRoC ships four (k3 = 0)
        }
        for (size_t k = 0; k < d.size(); ++k)
            camera.c.dist[k] = d[k];
        const size_t tbs = text.find("T_BS:");
        if (tbs == std::string::npos)
            throw std::runtime_error("readCameraFile: no key 'T_BS' in " + fileName);
        cameraOffset = poseFromRowMajor(numbersAfterKey(text, tbs, "data", 16, fileName));
    } else {
        camera.c.model = EQVIO_CAMERA_EQUIDISTANT;
        const std::vector<double> d = numbersAfterKey(text, from, "distortion_coeffs", 4, fileName);
        for (size_t k = 0; k < 4; ++k)
            camera.c.dist[k] = d[k];
        const Pose imuInCamera = poseFromRowMajor(numbersAfterKey(text, from, "T_cam_imu", 16, fileName));
        const Qt Ri = eqf::q_inv(imuInCamera.R);
        const V3 xi = eqf::q_rot(Ri, imuInCamera.x);
        cameraOffset = Pose{Ri, V3{-xi.x, -xi.y, -xi.z}};
    }
}

} // namespace eqvio_amd
