"""Host-side unit tests of the C++ filter mirror that need no device: compiled against eqvio_amd/host/ and run on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "eqvio_amd", "lib")


def build_and_run(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-I",
                    os.path.join(ROOT, "eqvio_amd", "host"), os.path.join(ROOT, "tests", "host", name + ".cpp"), "-L", LIBDIR, "-leqvio_filter", "-leqf_hip",
                    "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return subprocess.run([exe], check=True, capture_output=True, text=True).stdout


def test_vision_measurement_flat_cache_follows_the_map(tmp_path):
    """the cached flat arrays used to be trusted on size + first/last id; a reused measurement with new pixels or a swapped
    interior id went to the device stale. Every access now validates all ids and pixels."""
    assert build_and_run(tmp_path, "measurement_cache").strip() == "ok"
