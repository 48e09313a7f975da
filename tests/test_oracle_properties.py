"""The reference's gtest property tests (test/*.cpp), re-stated against the CPU oracle in
oracle/prop_tests.cpp. This is what pins the oracle: the reference has no golden vectors."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def prop_binary():
    subprocess.run(["make", "-s", "-C", ORACLE, "_prop_tests"], check=True)
    return os.path.join(ORACLE, "_prop_tests")


def _names():
    # keep in sync with oracle/prop_tests.cpp main(); listed statically so collection needs no build
    return [
        "VIOGroupTest.BasicOperations", "VIOActionTest.StateAction", "VIOActionTest.OutputAction", "VIOActionTest.OutputEquivariance",
        "VIOLiftTest.Lift", "VIOLiftTest.DiscreteLift", "VIOLiftTest.InnovationLifts_euclid", "VIOLiftTest.InnovationLifts_invdepth",
        "EqFMatricesTest.euclid_invdepth_compatibility",
        "EqFSuiteTest.stateMatrixA.euclid", "EqFSuiteTest.stateMatrixA.invdepth", "EqFSuiteTest.stateMatrixA.normal",
        "EqFSuiteTest.inputMatrixB.euclid", "EqFSuiteTest.inputMatrixB.invdepth", "EqFSuiteTest.inputMatrixB.normal",
        "EqFSuiteTest.outputMatrixC.euclid", "EqFSuiteTest.outputMatrixC.invdepth", "EqFSuiteTest.outputMatrixC.normal",
        "EqFSuiteTest.outputMatrixCStar",
        "CoordinateChartTest.SphereChartE3", "CoordinateChartTest.SphereChartPole", "CoordinateChartTest.SphereChartPoleNormal",
        "CoordinateChartTest.SphereChartE3Differential", "CoordinateChartTest.SphereChartPoleDifferential",
        "CoordinateChartTest.SphereChartPoleDifferentialNormal", "CoordinateChartTest.VIOChart_euclid", "CoordinateChartTest.VIOChart_invdepth",
        "CoordinateChartTest.VIOChart_normal", "CoordinateChartTest.VIOChart_euclid_invdepth_diff", "CoordinateChartTest.VIOChart_euclid_normal_diff",
        "FilterStatisticsTest.initialDistribution", "FilterStatisticsTest.trueInputDistribution", "FilterStatisticsTest.inputDistribution",
        "FilterStatisticsTest.outputDistribution", "Oracle.updateArithmeticsAgree", "Oracle.expmMatchesSeries",
    ]


def test_listed_names_match_binary(prop_binary):
    out = subprocess.run([prop_binary, "--list"], capture_output=True, text=True, check=True).stdout.split()
    assert out == _names()


@pytest.mark.parametrize("name", _names())
def test_reference_property(prop_binary, name):
    r = subprocess.run([prop_binary, name], capture_output=True, text=True)
    assert r.returncode == 0 and f"PASS {name}" in r.stdout, r.stdout[-2000:]
