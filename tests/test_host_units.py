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


def test_syrk_tile_order_is_a_permutation_and_xcd_compact():
    """k_syrk_sub's block -> tile table (host-side builder, no device): every lower tile exactly once for every tile count, and the blocks of one XCD
    (b % 8) touch far fewer 32-row panels of W than the plain column order did (which touched all of them on every XCD)."""
    import ctypes as C

    import numpy as np

    from eqvio_amd.capi import load_eqf_lib

    lib = load_eqf_lib()
    for nt in (1, 2, 3, 7, 8, 20, 33, 48, 64):
        ntiles = nt * (nt + 1) // 2
        out = np.full(ntiles, -1, np.int32)
        assert lib.eqf_debug_syrk_order(nt, out.ctypes.data_as(C.POINTER(C.c_int))) == 0
        bi, bj = out & 0xFFFF, out >> 16
        assert np.all((bj >= 0) & (bj <= bi) & (bi < nt))
        assert len({(int(a), int(b)) for a, b in zip(bi, bj)}) == ntiles
        if nt >= 20:
            panels = [len(set(bi[x::8].tolist()) | set(bj[x::8].tolist())) for x in range(8)]
            assert max(panels) <= 0.8 * nt and sum(panels) <= 5.5 * nt, (nt, panels)  # e.g. nt = 48: 18 .. 36 of 48 panels per XCD, 223 in all (plain order: 8 x 48)
            assert abs(len(bi[0::8]) - len(bi[7::8])) <= 1  # equal shares
