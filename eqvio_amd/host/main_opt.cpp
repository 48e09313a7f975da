// eqvio_opt: the reference's dataset main (src/main_opt.cpp:178-269) on the MI355X EqF path, fed by precomputed feature
// tracks instead of images (no OpenCV / GIFT here): IMU -> processIMUData, tracks -> processVisionData, outputs through
// VIOWriter. The filter starts uninitialised and sets its attitude from the first IMU sample (VIOFilter.cpp:65-78).
#include "DatasetReplay.hpp"
#include "VIOWriter.hpp"
#include "cli.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>

using namespace eqvio_amd;

static void usage() {
    std::puts("usage: eqvio_opt --imu FILE --features FILE [--format asl|uzhfpv] [--groundtruth FILE] [--dumpMeasurements] [--dumpStates FILE] [--printCamera]\n"
              "                 [--cameraFile sensor.yaml | camchain.yaml]   (intrinsics, distortion and camera offset from the dataset's own file, main_opt.cpp:114-147)\n"
              "                 [--camera fx fy cx cy width height] [--distortion radtan k1 k2 p1 p2 k3 | --distortion equidistant k1 k2 k3 k4]\n"
              "                 [--cameraOffset qw qx qy qz x y z] [--cameraLag S] [--start S] [--stop S] [--output DIR] [--sigmaFP32] [--quiet]\n"
              "                 [--<eqf setting> VALUE ...]   (names of VIOFilter::Settings, e.g. --fastRiccati 1 --coordinateChoice InvDepth)");
}

int main(int argc, char** argv) {
    VIOFilter::Settings fs;
    std::string imuName, featName, gtName, outputDir, cameraFileName, statesName;
    DatasetFormat format = DatasetFormat::ASL;
    auto cam = std::make_shared<Camera>();
    cam->c.fx = 458.654; // intrinsics.yaml:7 (EuRoC cam0)
    cam->c.fy = 457.296;
    cam->c.cx = 367.215;
    cam->c.cy = 248.375;
    cam->c.width = 752;
    cam->c.height = 480;
    double cameraLag = 0, startTime = -1, stopTime = -1;
    bool quiet = false, dump = false, sigmaFP32 = false, printCamera = false;
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            std::function<const char*()> val = [&]() -> const char* {
                if (i + 1 >= argc)
                    throw std::runtime_error("missing value after " + a);
                return argv[++i];
            };
            if (a == "--imu") imuName = val();
            else if (a == "--features") featName = val();
            else if (a == "--groundtruth") gtName = val();
            else if (a == "--cameraFile") cameraFileName = val();
            else if (a == "--dumpStates") statesName = val();
            else if (a == "--format") {
                const std::string f = val();
                if (f == "asl") format = DatasetFormat::ASL;
                else if (f == "uzhfpv") format = DatasetFormat::UZHFPV;
                else throw std::runtime_error("unknown --format " + f);
            } else if (a == "--camera") {
                cam->c.fx = std::atof(val());
                cam->c.fy = std::atof(val());
                cam->c.cx = std::atof(val());
                cam->c.cy = std::atof(val());
                cam->c.width = std::atoi(val());
                cam->c.height = std::atoi(val());
            } else if (a == "--distortion") { // radtan k1 k2 p1 p2 k3 (sensor.yaml distortion_coefficients) | equidistant k1 k2 k3 k4
                const std::string mdl = val();
                if (mdl == "radtan") {
                    cam->c.model = EQVIO_CAMERA_RADTAN;
                    for (int k = 0; k < 5; ++k)
                        cam->c.dist[k] = std::atof(val());
                } else if (mdl == "equidistant") {
                    cam->c.model = EQVIO_CAMERA_EQUIDISTANT;
                    for (int k = 0; k < 4; ++k)
                        cam->c.dist[k] = std::atof(val());
                } else
                    throw std::runtime_error("unknown --distortion model " + mdl + " (radtan | equidistant)");
            } else if (a == "--cameraOffset") {
                double q[7];
                for (double& v : q)
                    v = std::atof(val());
                fs.cameraOffset = Pose{eqf::q_unit(Qt{q[0], q[1], q[2], q[3]}), V3{q[4], q[5], q[6]}};
            } else if (a == "--cameraLag") cameraLag = std::atof(val());
            else if (a == "--start") startTime = std::atof(val());
            else if (a == "--stop") stopTime = std::atof(val());
            else if (a == "--output") outputDir = val();
            else if (a == "--quiet") quiet = true;
            else if (a == "--sigmaFP32") sigmaFP32 = true;
            else if (a == "--dumpMeasurements") dump = true;
            else if (a == "--printCamera") printCamera = true;
            else if (!parseFilterFlag(a, val, fs)) {
                usage();
                return a == "--help" ? 0 : 2;
            }
        }
        if (imuName.empty() || featName.empty()) {
            usage();
            return 2;
        }
        if (!cameraFileName.empty()) // after the flags: --format decides which layout the file has (main_opt.cpp:114-147)
            readCameraFile(cameraFileName, format, *cam, fs.cameraOffset);
        if (printCamera) { // host-only: the camera and the camera offset as the run would use them (model fx fy cx cy width height k1..k5 | qw qx qy qz x y z)
            std::printf("camera %d %.17g %.17g %.17g %.17g %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", cam->c.model, cam->c.fx, cam->c.fy, cam->c.cx,
                        cam->c.cy, cam->c.width, cam->c.height, cam->c.dist[0], cam->c.dist[1], cam->c.dist[2], cam->c.dist[3], cam->c.dist[4], fs.cameraOffset.R.w, fs.cameraOffset.R.x,
                        fs.cameraOffset.R.y, fs.cameraOffset.R.z, fs.cameraOffset.x.x, fs.cameraOffset.x.y, fs.cameraOffset.x.z);
            return 0;
        }
        TrackReplayServer dataServer(imuName, featName, format, cam, cameraLag);
        if (dump) { // host-only: print the merged measurement stream as parsed (no filter, no device)
            std::printf("%s", "");
            while (dataServer.nextMeasurementType() != MeasurementType::None) {
                const double t = dataServer.nextTime();
                if (dataServer.nextMeasurementType() == MeasurementType::IMU) {
                    const IMUVelocity v = dataServer.getIMU();
                    std::printf("IMU %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", v.stamp, v.gyr.x, v.gyr.y, v.gyr.z, v.acc.x, v.acc.y, v.acc.z);
                    if (v.stamp != t)
                        throw std::runtime_error("nextTime disagrees with the IMU stamp");
                } else {
                    const VisionMeasurement m = dataServer.getSimVision();
                    std::printf("IMG %.17g %zu", m.stamp, m.camCoordinates.size());
                    for (const auto& kv : m.camCoordinates)
                        std::printf(" %d %.17g %.17g", kv.first, kv.second[0], kv.second[1]);
                    std::printf("\n");
                }
            }
            if (!gtName.empty()) {
                const std::vector<StampedPose> gt = TrackReplayServer::groundtruth(gtName, format);
                std::printf("GT %zu", gt.size());
                if (!gt.empty())
                    std::printf(" %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", gt[0].t, gt[0].pose.x.x, gt[0].pose.x.y, gt[0].pose.x.z, gt[0].pose.R.w, gt[0].pose.R.x,
                                gt[0].pose.R.y, gt[0].pose.R.z);
                std::printf("\n");
            }
            return 0;
        }
        loopTimer.initialise({"correction", "features", "preprocessing", "propagation", "total", "total vision update", "write output"});
        VIOFilter filter(fs); // main_opt.cpp:150
        if (sigmaFP32) // BASELINE config 5: Sigma stored as float in HBM (include/eqf_hip.h)
            eqf_set_option(filter.eqfState().ctx, EQF_OPT_SIGMA_FP32, 2);
        std::unique_ptr<VIOWriter> vioWriter;
        if (!outputDir.empty())
            vioWriter = std::make_unique<VIOWriter>(outputDir);
        // --dumpStates: the state estimate after every vision measurement at FULL precision (%.17g; the writer's files carry 6 digits), one line per frame:
        // time, the 23 numbers of the sensor state (bias, pose wxyz + xyz, velocity, camera offset), N, then id x y z per landmark. For parity tests.
        std::FILE* statesFile = statesName.empty() ? nullptr : std::fopen(statesName.c_str(), "w");
        if (!statesName.empty() && !statesFile)
            throw std::runtime_error("cannot open " + statesName);
        int imuDataCounter = 0, visionDataCounter = 0;
        const auto loopStartTime = std::chrono::steady_clock::now();
        while (true) {
            const MeasurementType measType = dataServer.nextMeasurementType();
            if (measType == MeasurementType::None)
                break;
            if (measType == MeasurementType::Image) {
                loopTimer.startLoop();
                loopTimer.startTiming("total");
                VisionMeasurement measData = dataServer.getSimVision();
                if (startTime > 0 && measData.stamp < startTime)
                    continue;
                loopTimer.startTiming("total vision update");
                filter.processVisionData(measData);
                loopTimer.endTiming("total vision update");
                loopTimer.endTiming("total");
                ++visionDataCounter;
                loopTimer.startTiming("write output");
                const VIOState estimatedState = filter.stateEstimate();
                if (statesFile) {
                    const VIOSensorState& se = estimatedState.sensor;
                    std::fprintf(statesFile, "%.17g", filter.getTime());
                    const double sv[23] = {se.inputBias[0], se.inputBias[1], se.inputBias[2], se.inputBias[3], se.inputBias[4], se.inputBias[5], se.pose.R.w, se.pose.R.x, se.pose.R.y,
                                           se.pose.R.z, se.pose.x.x, se.pose.x.y, se.pose.x.z, se.velocity.x, se.velocity.y, se.velocity.z, se.cameraOffset.R.w, se.cameraOffset.R.x,
                                           se.cameraOffset.R.y, se.cameraOffset.R.z, se.cameraOffset.x.x, se.cameraOffset.x.y, se.cameraOffset.x.z};
                    for (const double v : sv)
                        std::fprintf(statesFile, " %.17g", v);
                    std::fprintf(statesFile, " %zu", estimatedState.cameraLandmarks.size());
                    for (const Landmark& lm : estimatedState.cameraLandmarks)
                        std::fprintf(statesFile, " %d %.17g %.17g %.17g", lm.id, lm.p.x, lm.p.y, lm.p.z);
                    std::fprintf(statesFile, "\n");
                }
                if (vioWriter) {
                    vioWriter->writeStates(filter.getTime(), estimatedState);
                    vioWriter->writeFeatures(measData);
                }
                loopTimer.endTiming("write output");
                if (vioWriter)
                    vioWriter->writeTiming(loopTimer.getLoopTimingData());
            } else {
                const IMUVelocity imuData = dataServer.getIMU();
                if (startTime > 0 && imuData.stamp < startTime)
                    continue;
                filter.processIMUData(imuData);
                ++imuDataCounter;
            }
            if (stopTime > 0 && filter.getTime() > stopTime)
                break;
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStartTime).count();
        if (statesFile)
            std::fclose(statesFile);
        std::cout << "Processed " << imuDataCounter << " IMU and " << visionDataCounter << " vision measurements.\n"
                  << "Time taken: " << elapsed << " seconds." << std::endl;
        const VIOState est = filter.stateEstimate();
        std::printf("final time %.9g  position %.6g %.6g %.6g  landmarks %d  vision updates/s %.1f\n", filter.getTime(), est.sensor.pose.x.x, est.sensor.pose.x.y,
                    est.sensor.pose.x.z, filter.viewEqFState().numLandmarks(), visionDataCounter / elapsed);
        if (!gtName.empty()) { // distance to the ground-truth pose nearest in time, after aligning the first poses
            const std::vector<StampedPose> gt = TrackReplayServer::groundtruth(gtName, format);
            if (!gt.empty()) {
                size_t k = 0;
                for (size_t j = 0; j < gt.size(); ++j)
                    if (std::fabs(gt[j].t - filter.getTime()) < std::fabs(gt[k].t - filter.getTime()))
                        k = j;
                std::printf("groundtruth poses %zu  nearest stamp %.9g  position %.6g %.6g %.6g\n", gt.size(), gt[k].t, gt[k].pose.x.x, gt[k].pose.x.y, gt[k].pose.x.z);
            }
        }
        (void)quiet;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "eqvio_opt: %s\n", e.what());
        return 1;
    }
    return 0;
}
