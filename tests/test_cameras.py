"""Camera models of the EqF path (include/eqvio_types.h): pinhole, radial-tangential (GIFT::StandardCamera, EuRoC) and
equidistant (GIFT::EquidistantCamera, UZH-FPV). GIFT is an un-vendored submodule of the reference, so the models are
pinned here by their defining properties (inverse pair, Jacobian = derivative, pinhole limit), by the reference's own
EuRoC coefficients (intrinsics.yaml:7-8), and by agreement between two independent implementations: the oracle
(fixed-point + Newton inverse, oracle/vio.hpp) and the product's shared host/device math (Newton, eqf_math.hpp).
CPU tests run the product math on the host; the gpu tests run it in the kernels (C matrix, update, outlier statistics)."""
import numpy as np
import pytest

from eqvio_amd.capi import Camera
from oracle_binding import oracle_cam_jacobian, oracle_cam_project, oracle_cam_undistort
from util import CAMERAS, CHARTS

IMPLS = {"oracle": (oracle_cam_project, oracle_cam_undistort, oracle_cam_jacobian),
         "product": (lambda c, p: c.project(p), lambda c, y: c.undistort(y), lambda c, p: c.jacobian(p))}


def pixel_grid(cam, k=9, margin=0.0):
    us = np.linspace(margin, cam.width - 1 - margin, k)
    vs = np.linspace(margin, cam.height - 1 - margin, k)
    return [np.array([u, v]) for u in us for v in vs]


@pytest.mark.parametrize("impl", list(IMPLS))
@pytest.mark.parametrize("model", list(CAMERAS))
def test_undistort_inverts_project_over_the_whole_image(model, impl):
    project, undistort, _ = IMPLS[impl]
    cam = CAMERAS[model]()
    for y in pixel_grid(cam):
        b = undistort(cam, y)
        assert abs(np.linalg.norm(b) - 1.0) < 1e-14 and b[2] > 0
        for depth in (0.3, 7.0):
            np.testing.assert_allclose(project(cam, depth * b), y, rtol=0, atol=2e-10)


@pytest.mark.parametrize("impl", list(IMPLS))
@pytest.mark.parametrize("model", list(CAMERAS))
def test_jacobian_is_the_derivative_of_project(model, impl):
    project, undistort, jacobian = IMPLS[impl]
    cam = CAMERAS[model]()
    rng = np.random.default_rng(0)
    for y in pixel_grid(cam, k=5):
        p = rng.uniform(1.0, 9.0) * undistort(cam, y)
        J = jacobian(cam, p)
        h = 1e-6
        num = np.stack([(project(cam, p + h * e) - project(cam, p - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
        np.testing.assert_allclose(J, num, rtol=0, atol=2e-6 * max(1.0, np.max(np.abs(J))))
        np.testing.assert_allclose(J @ p, 0.0, atol=1e-9 * np.max(np.abs(J)) * np.linalg.norm(p))  # projection is scale invariant


@pytest.mark.parametrize("model", list(CAMERAS))
def test_oracle_and_product_agree(model):
    cam = CAMERAS[model]()
    rng = np.random.default_rng(1)
    for y in pixel_grid(cam, k=11):
        bo, bp = oracle_cam_undistort(cam, y), cam.undistort(y)
        np.testing.assert_allclose(bp, bo, rtol=0, atol=1e-13)
        p = rng.uniform(0.5, 20.0) * bo
        np.testing.assert_allclose(cam.project(p), oracle_cam_project(cam, p), rtol=0, atol=1e-10)
        Jo = oracle_cam_jacobian(cam, p)
        np.testing.assert_allclose(cam.jacobian(p), Jo, rtol=0, atol=1e-11 * max(1.0, np.max(np.abs(Jo))))


def test_zero_distortion_is_the_pinhole():
    pin = Camera.pinhole(458.654, 457.296, 367.215, 248.375, 752, 480)
    for other in (Camera.radtan(458.654, 457.296, 367.215, 248.375, 752, 480, 0, 0, 0, 0, 0),):
        for y in pixel_grid(pin, k=4):
            np.testing.assert_allclose(other.undistort(y), pin.undistort(y), rtol=0, atol=1e-15)
            p = 3.0 * pin.undistort(y)
            np.testing.assert_allclose(other.project(p), pin.project(p), rtol=0, atol=1e-12)
            np.testing.assert_allclose(oracle_cam_project(other, p), oracle_cam_project(pin, p), rtol=0, atol=1e-12)


def test_euroc_distortion_known_values():
    """Known-answer check of the radial-tangential model with the reference's EuRoC coefficients (intrinsics.yaml:8),
    evaluated by hand from the OpenCV definition at the normalised point (0.3, -0.2)."""
    cam = CAMERAS["radtan"]()
    x, y = 0.3, -0.2
    k1, k2, p1, p2 = -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    expect = np.array([458.654 * xd + 367.215, 457.296 * yd + 248.375])
    for project in (oracle_cam_project, lambda c, p: c.project(p)):
        np.testing.assert_allclose(project(cam, np.array([x, y, 1.0]) * 2.5), expect, rtol=0, atol=1e-10)
    assert abs(expect[0] - 367.215 - 458.654 * x) > 1.0  # the distortion really moves the pixel


# --------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("model", ["radtan", "equidistant"])
@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("star", [True, False])
def test_output_matrix_and_update_with_distorted_cameras(model, chart, star):
    from test_gpu_parity import check_sigma, check_state, make_pair
    from util import settings_for, synth_measurement

    cam = CAMERAS[model]()
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], 23, seed=41, useEquivariantOutput=int(star), useDiscreteInnovationLift=0)
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.5, subset=rng.permutation(23)[:17])
    C_g, yt_g = core.debug_matrix_C(cam, mid, y, star)
    C_o = orc.output_matrix_C(cam, mid, y, star)
    assert np.max(np.abs(C_g - C_o)) <= 1e-10 * max(1.0, np.max(np.abs(C_o)))
    a_g, p_g, d_g = core.outlier_stats(cam, mid, y)
    a_o, p_o = orc.outlier_stats(cam, mid, y)[:2]
    np.testing.assert_allclose(a_g, a_o, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(p_g, p_o, rtol=1e-8, atol=1e-9)
    orc.vision_update(cam, mid, y)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, star, False)
    check_sigma(core, orc)
    check_state(core, orc)
