// See DatasetReplay.hpp.
#include "DatasetReplay.hpp"
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

} // namespace eqvio_amd
