// eqvio_sim: the reference's simulation main (src/main_sim.cpp:128-184) on the MI355X EqF path.
// Same loop: Image -> [augmentLandmarkStates] -> processVisionData -> stateEstimate / computeNEES / write;
// IMU -> processIMUData -> optional landmark reset. Configuration comes from --key value flags instead of a YAML file
// (yaml-cpp and argparse are not in this image); defaults are the reference's.
#include "VIOSimulator.hpp"
#include "VIOWriter.hpp"
#include "cli.hpp"
#include <fstream>
#include <iomanip>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace eqvio_amd;

static void usage() {
    std::puts("usage: eqvio_sim [--duration S] [--trajectory wave|square|sine|line] [--numPoints N] [--numWalls W] [--wallDistance D]\n"
              "                 [--maxFeatures M] [--seed S] [--imuFreq HZ] [--imageFreq HZ] [--initialNoise] [--inputNoise] [--outputNoise]\n"
              "                 [--fullState] [--landmarkReset S] [--output DIR] [--writeDataset DIR] [--sigmaFP32] [--quiet]\n"
              "                 [--<eqf setting> VALUE ...]   (names of VIOFilter::Settings, e.g. --fastRiccati 1 --coordinateChoice InvDepth)");
}

int main(int argc, char** argv) {
    SimSettings sim;
    sim.duration = 20.0;
    VIOFilter::Settings fs;
    bool fullState = false, quiet = false, sigmaFP32 = false;
    double landmarkResetTime = -1.0;
    std::string outputDir, datasetDir;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        std::function<const char*()> val = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage();
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--duration") sim.duration = std::atof(val());
        else if (a == "--trajectory") sim.trajectory = val();
        else if (a == "--numPoints") sim.numPoints = std::atoi(val());
        else if (a == "--numWalls") sim.numWalls = std::atoi(val());
        else if (a == "--wallDistance") sim.wallDistance = std::atof(val());
        else if (a == "--maxFeatures") sim.maxFeatures = (size_t)std::atoi(val());
        else if (a == "--seed") sim.randomSeed = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--imuFreq") sim.imuFreq = std::atof(val());
        else if (a == "--imageFreq") sim.imageFreq = std::atof(val());
        else if (a == "--initialNoise") sim.initialNoise = true;
        else if (a == "--inputNoise") sim.inputNoise = true;
        else if (a == "--outputNoise") sim.outputNoise = true;
        else if (a == "--fullState") fullState = true;
        else if (a == "--landmarkReset") landmarkResetTime = std::atof(val());
        else if (a == "--output") outputDir = val();
        else if (a == "--writeDataset") datasetDir = val();
        else if (a == "--quiet") quiet = true;
        else if (a == "--sigmaFP32") sigmaFP32 = true;
        else if (parseFilterFlag(a, val, fs)) {
        } else {
            usage();
            return a == "--help" ? 0 : 2;
        }
    }
    double lastLandmarkReset = landmarkResetTime > 0 ? 0.0 : std::nan("");

    SimulationDataServer simDataServer(sim, fs);
    loopTimer.initialise({"correction", "features", "preprocessing", "propagation", "total", "total vision update", "write output"});
    // camera extrinsics provided by the data server override the filter settings (main_sim.cpp:97-101)
    fs.cameraOffset = *simDataServer.cameraExtrinsics();
    // the initial condition carries ALL world points (main_sim.cpp:105); the first augmentLandmarkStates trims it
    fs.maxLandmarks = std::max(fs.maxLandmarks, sim.numPoints + (int)sim.maxFeatures);

    std::unique_ptr<VIOWriter> vioWriter; // main_sim.cpp:108-122 (writeState)
    if (!outputDir.empty())
        vioWriter = std::make_unique<VIOWriter>(outputDir);

    // --writeDataset DIR: the measurements of this run in the ASL layout eqvio_opt reads (imu.csv with ns stamps; the
    // feature tracks are features.csv of --output), plus the ground-truth poses
    std::ofstream imuOut, gtOut;
    if (!datasetDir.empty()) {
        if (datasetDir.back() != '/')
            datasetDir += '/';
        VIOWriter makeDir(datasetDir);
        imuOut.open(datasetDir + "imu.csv");
        imuOut << "#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n";
        gtOut.open(datasetDir + "groundtruth.csv");
        gtOut << "#timestamp [ns],p_RS_R_x [m],p_RS_R_y [m],p_RS_R_z [m],q_RS_w [],q_RS_x [],q_RS_y [],q_RS_z []\n";
    }

    try {
        VIOFilter filter(simDataServer.getInitialCondition(), fs);
        if (sigmaFP32) // BASELINE config 5: Sigma stored as float in HBM (include/eqf_hip.h)
            eqf_set_option(filter.eqfState().ctx, EQF_OPT_SIGMA_FP32, 2);
        int imuDataCounter = 0, visionDataCounter = 0;
        double neesSum = 0, neesMax = 0, posErr = 0;
        int neesFailures = 0;
        const auto loopStartTime = std::chrono::steady_clock::now();
        if (!quiet)
            std::cout << "NEES:\n";
        while (true) {
            const MeasurementType measType = simDataServer.nextMeasurementType();
            if (measType == MeasurementType::None)
                break;
            if (measType == MeasurementType::Image) {
                loopTimer.startLoop(); // as in main_opt.cpp:180: timing.csv gets one row per vision frame
                VisionMeasurement measData = simDataServer.getSimVision();
                if (!fullState)
                    filter.augmentLandmarkStates(measData.getIds(), simDataServer.getTrueState(measData.stamp, true));
                filter.processVisionData(measData);
                ++visionDataCounter;
                const VIOState estimatedState = filter.stateEstimate();
                const VIOState trueState = simDataServer.getTrueState(filter.getTime());
                double NEES = std::nan("");
                try {
                    NEES = filter.viewEqFState().computeNEES(trueState);
                } catch (const std::exception&) {
                    // Sigma is factorised on the device (Cholesky-type); a Sigma that is positive definite only up to rounding
                    // (cond > 1e13, e.g. the template's 0.003 px measurement noise) is reported as NaN here, the run goes on
                    ++neesFailures;
                }
                if (NEES == NEES) {
                    neesSum += NEES;
                    neesMax = std::max(neesMax, NEES);
                }
                posErr = eqf::norm(estimatedState.sensor.pose.x - trueState.sensor.pose.x);
                if (vioWriter) { // main_sim.cpp:149-154
                    vioWriter->writeStates(filter.getTime(), estimatedState);
                    vioWriter->writeFeatures(measData);
                    vioWriter->writeLandmarkError(filter.getTime(), trueState, estimatedState);
                    vioWriter->writeConsistency(filter.getTime(), trueState, filter.viewEqFState());
                    vioWriter->writeTiming(loopTimer.getLoopTimingData());
                }
                if (!quiet)
                    std::cout << '\r' << NEES << std::flush;
            } else {
                const IMUVelocity imuData = simDataServer.getIMU();
                if (imuOut.is_open()) {
                    imuOut << std::llround(imuData.stamp * 1e9) << std::setprecision(17) << ',' << imuData.gyr.x << ',' << imuData.gyr.y << ',' << imuData.gyr.z << ','
                           << imuData.acc.x << ',' << imuData.acc.y << ',' << imuData.acc.z << '\n';
                    const VIOState t = simDataServer.getTrueState(imuData.stamp);
                    gtOut << std::llround(imuData.stamp * 1e9) << std::setprecision(17) << ',' << t.sensor.pose.x.x << ',' << t.sensor.pose.x.y << ',' << t.sensor.pose.x.z << ','
                          << t.sensor.pose.R.w << ',' << t.sensor.pose.R.x << ',' << t.sensor.pose.R.y << ',' << t.sensor.pose.R.z << '\n';
                }
                filter.processIMUData(imuData);
                ++imuDataCounter;
                if (filter.getTime() >= lastLandmarkReset + landmarkResetTime) { // false while lastLandmarkReset is NaN
                    lastLandmarkReset += landmarkResetTime;
                    filter.setLandmarks(simDataServer.getTrueState(filter.getTime(), true).cameraLandmarks);
                }
            }
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStartTime).count();
        std::cout << "\n\nProcessed " << imuDataCounter << " IMU and " << visionDataCounter << " vision measurements.\n"
                  << "Time taken: " << elapsed << " seconds." << std::endl;
        std::printf("mean NEES %.6g  max NEES %.6g  final position error %.6g m  landmarks %d  vision updates/s %.1f\n", neesSum / std::max(visionDataCounter - neesFailures, 1), neesMax,
                    posErr, filter.viewEqFState().numLandmarks(), visionDataCounter / elapsed);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "eqvio_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
