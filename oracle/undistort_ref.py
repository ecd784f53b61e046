"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame undistortion of
pi3/utils/undistortion.py (map builder :95-138, cv2.remap :157-177), SURVEY.md §8f rank 4.

PARITY UNPINNED: the arithmetic lives in two third-party libraries that are neither vendored in /root/reference nor
installed here - pytheia 0.2.9 (TheiaSfM camera models: ImageToCameraCoordinates / CameraToImageCoordinates) and OpenCV
(cv2.remap, INTER_LINEAR, 8-bit).  Both are restated from their published sources (TheiaSfM
src/theia/sfm/camera/*_camera_model.h; OpenCV modules/imgproc/src/imgwarp.cpp: remap -> remapBilinear, INTER_BITS = 5,
INTER_REMAP_COEF_BITS = 15) and anchored on the reference's call sites; the tests are known-answer and property tests.
"""
from __future__ import annotations

import numpy as np


def project(model: str, x, y, f, ar, cx, cy, skew, radial, tangential):
    """CameraToImageCoordinates of the ray (x, y, 1) for the four TheiaSfM models used by pi3/utils/camera.py."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if model == "DIVISION_UNDISTORTION":
        ux, uy = f * x, f * ar * y
        rr, k = ux * ux + uy * uy, radial[0]
        sc = np.ones_like(rr)
        if abs(k) >= 1e-15:
            inner = 1.0 - 4.0 * k * rr
            ok = (rr >= 1e-15) & (inner >= 0.0)
            with np.errstate(invalid="ignore", divide="ignore"):
                sc = np.where(ok, (1.0 - np.sqrt(np.where(ok, inner, 1.0))) / np.where(ok, 2.0 * k * rr, 1.0), 1.0)
        return ux * sc + cx, uy * sc + cy
    r_sq = x * x + y * y
    if model == "FISHEYE":
        rad = np.sqrt(r_sq)
        th = np.arctan2(rad, 1.0)
        th2 = th * th
        thd = th * (1.0 + th2 * (radial[0] + th2 * (radial[1] + th2 * (radial[2] + th2 * radial[3]))))
        small = r_sq < 1e-8
        safe = np.where(small, 1.0, rad)
        dx, dy = np.where(small, x, thd * x / safe), np.where(small, y, thd * y / safe)
    else:
        k3 = radial[2] if model == "PINHOLE_RADIAL_TANGENTIAL" else 0.0
        radial_f = 1.0 + r_sq * (radial[0] + r_sq * (radial[1] + r_sq * k3))
        dx, dy = x * radial_f, y * radial_f
        if model == "PINHOLE_RADIAL_TANGENTIAL":
            xy = x * y
            dx = dx + 2.0 * tangential[0] * xy + tangential[1] * (r_sq + 2.0 * x * x)
            dy = dy + tangential[0] * (r_sq + 2.0 * y * y) + 2.0 * tangential[1] * xy
    return f * dx + skew * dy + cx, f * ar * dy + cy


def undistort_maps(calib: dict, target_hw):
    """UndistortionMaps.compute_maps for the automatically created undistorted camera (scale 1.0), vectorised."""
    intr, model = calib["intrinsics"], calib["intrinsic_type"]
    f, ar = float(intr["focal_length"]), float(intr["aspect_ratio"])
    cx, cy, skew = float(intr["principal_pt_x"]), float(intr["principal_pt_y"]), float(intr.get("skew", 0.0))
    if model == "DIVISION_UNDISTORTION":
        radial, tang = [float(intr["div_undist_distortion"]), 0, 0, 0], [0.0, 0.0]
    elif model == "FISHEYE":
        radial, tang = [float(intr[f"radial_distortion_{i}"]) for i in (1, 2, 3, 4)], [0.0, 0.0]
    elif model == "PINHOLE":
        radial, tang = [float(intr["radial_distortion_1"]), float(intr["radial_distortion_2"]), 0, 0], [0.0, 0.0]
    else:
        radial = [float(intr[f"radial_distortion_{i}"]) for i in (1, 2, 3)] + [0.0]
        tang = [float(intr["tangential_distortion_1"]), float(intr["tangential_distortion_2"])]
    H, W = target_hw
    c, r = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    # undistorted camera: same focal length / principal point / skew, aspect ratio 1, zero distortion
    y = (r - cy) / (f * 1.0)
    x = (c - cx) / f if model == "DIVISION_UNDISTORTION" else (c - cx - skew * y) / f
    u, v = project(model, x, y, f, ar, cx, cy, skew, radial, tang)
    return u.astype(np.float32), v.astype(np.float32)


def remap_bilinear_u8(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(img, map_x, map_y, cv2.INTER_LINEAR) for uint8 HWC, BORDER_CONSTANT 0 (pure numpy, vectorised)."""
    H0, W0 = img.shape[:2]
    with np.errstate(invalid="ignore"):
        qx = np.nan_to_num(np.rint(map_x.astype(np.float32) * np.float32(32.0)), nan=0.0)
        qy = np.nan_to_num(np.rint(map_y.astype(np.float32) * np.float32(32.0)), nan=0.0)
    qx = np.clip(qx, -2147483648.0, 2147483520.0).astype(np.int64)
    qy = np.clip(qy, -2147483648.0, 2147483520.0).astype(np.int64)
    sx, sy = np.clip(qx >> 5, -32768, 32767), np.clip(qy >> 5, -32768, 32767)
    fx, fy = qx & 31, qy & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.full(map_x.shape + (img.shape[2],), 1 << 14, dtype=np.int64)
    src = img.astype(np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H0) & (xx >= 0) & (xx < W0)
        tap = src[np.clip(yy, 0, H0 - 1), np.clip(xx, 0, W0 - 1)]
        acc += np.where(ok[..., None], tap, 0) * w[k][..., None]
    return np.minimum(acc >> 15, 255).astype(np.uint8)


def undistort_frames(frames_u8: np.ndarray, calib: dict, target_hw) -> np.ndarray:
    """uint8 [N,H0,W0,3] -> float32 [N,3,H,W]: maps, remap, ToTensor (datasets/image_datasets.py:192-199)."""
    mx, my = undistort_maps(calib, target_hw)
    out = np.stack([remap_bilinear_u8(f, mx, my) for f in frames_u8])
    return (out.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2).copy()
