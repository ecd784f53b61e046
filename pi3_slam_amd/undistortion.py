"""Frame undistortion on the device: mirror of pi3/utils/camera.py (calibration JSON), pi3/utils/undistortion.py
(UndistortionMaps) and utils/undistortion_utils.py (create_undistortion_maps) for image-file inputs.

The reference builds its maps with pytheia camera models in a python double loop and applies them with cv2.remap; here
both steps are HIP kernels (csrc/undistort.hip).  Quirks kept on purpose:
  * the undistorted camera = the distorted one with distortion zeroed and aspect ratio 1; the reference also tries to
    re-centre the principal point but writes keys ("principal_point_x/y") that its loader never reads
    ("principal_pt_x/y"), so the principal point stays the calibrated one (undistortion.py:83-84 vs camera.py:82-83);
  * the maps are evaluated at the TARGET-size pixel grid with the native-scale intrinsics (scale = 1.0,
    undistortion_utils.py:31), so the result is the top-left target-size window of the undistorted native image, not a
    resized image (undistortion.py:118-132).
Video inputs and a frame size different from the calibration (the reference cv2.resize's those first) are out of scope.
"""
from __future__ import annotations

import json
from typing import Dict, Optional, Tuple

import torch

from . import ops

MODEL_IDS = {"PINHOLE": 0, "PINHOLE_RADIAL_TANGENTIAL": 1, "FISHEYE": 2, "DIVISION_UNDISTORTION": 3}


class Camera:
    """Calibration record with the reference's JSON schema (pi3/utils/camera.py:38-112)."""

    def __init__(self) -> None:
        self.cam_intr_json: Optional[Dict] = None
        self.scale = 1.0

    def load_camera_calibration_json(self, js: Dict, scale: float = 1.0) -> None:
        self.cam_intr_json = js
        self._load_camera_calibration(scale)

    def load_camera_calibration_file(self, path_to_json: str, scale: float = 1.0) -> None:
        with open(path_to_json, "r") as f:
            self.cam_intr_json = json.load(f)
        self._load_camera_calibration(scale)

    def _load_camera_calibration(self, scale: float = 1.0) -> None:
        js, intr = self.cam_intr_json, self.cam_intr_json["intrinsics"]
        model = js["intrinsic_type"]
        if model not in MODEL_IDS:
            raise ValueError(f"unsupported intrinsic_type {model!r}")
        self.model = model
        self.scale = scale
        self.image_width = int(js["image_width"] * scale)
        self.image_height = int(js["image_height"] * scale)
        self.focal_length = float(intr["focal_length"]) * scale
        self.aspect_ratio = float(intr["aspect_ratio"])
        self.principal_point = (float(intr["principal_pt_x"]) * scale, float(intr["principal_pt_y"]) * scale)
        self.skew = float(intr.get("skew", 0.0))
        if model == "DIVISION_UNDISTORTION":
            self.radial = [float(intr["div_undist_distortion"]), 0.0, 0.0, 0.0]
            self.tangential = [0.0, 0.0]
        elif model == "FISHEYE":
            self.radial = [float(intr[f"radial_distortion_{i}"]) for i in (1, 2, 3, 4)]
            self.tangential = [0.0, 0.0]
        elif model == "PINHOLE":
            self.radial = [float(intr["radial_distortion_1"]), float(intr["radial_distortion_2"]), 0.0, 0.0]
            self.tangential = [0.0, 0.0]
        else:
            self.radial = [float(intr[f"radial_distortion_{i}"]) for i in (1, 2, 3)] + [0.0]
            self.tangential = [float(intr["tangential_distortion_1"]), float(intr["tangential_distortion_2"])]


def undistorted_copy(cam: Camera) -> Camera:
    """UndistortionMaps._create_undistorted_camera (undistortion.py:51-93), including its principal-point quirk."""
    und = Camera()
    und.cam_intr_json = cam.cam_intr_json
    und._load_camera_calibration(cam.scale)
    und.radial, und.tangential = [0.0, 0.0, 0.0, 0.0], [0.0, 0.0]
    und.aspect_ratio = 1.0
    return und


class UndistortionMaps:
    def __init__(self, cam_dist: Camera, cam_undist: Optional[Camera] = None, device: str = "cuda"):
        self.cam_dist = cam_dist
        self.cam_undist = cam_undist if cam_undist is not None else undistorted_copy(cam_dist)
        self.device = device
        self.map_x = self.map_y = None
        self.current_target_size: Optional[Tuple[int, int]] = None

    def params16(self):
        u, d = self.cam_undist, self.cam_dist
        return [u.focal_length, u.aspect_ratio, u.principal_point[0], u.principal_point[1], u.skew,
                d.focal_length, d.aspect_ratio, d.principal_point[0], d.principal_point[1], d.skew,
                *d.radial, *d.tangential]

    def compute_maps(self, target_size: Optional[Tuple[int, int]] = None):
        H, W = target_size if target_size is not None else (self.cam_undist.image_height, self.cam_undist.image_width)
        self.map_x, self.map_y = ops.undistort_maps(self.params16(), MODEL_IDS[self.cam_dist.model], int(H), int(W),
                                                    self.device)
        self.current_target_size = target_size
        return self.map_x, self.map_y

    def get_maps(self, target_size: Optional[Tuple[int, int]] = None):
        if self.map_x is None or target_size != self.current_target_size:
            return self.compute_maps(target_size)
        return self.map_x, self.map_y

    def undistort_frames_device(self, frames_u8: torch.Tensor, target_size: Optional[Tuple[int, int]] = None):
        """uint8 [N,H0,W0,3] on the device -> float32 [N,3,H,W] in [0,1] (remap + ToTensor)."""
        if tuple(frames_u8.shape[1:3]) != (self.cam_dist.image_height, self.cam_dist.image_width):
            raise NotImplementedError(
                f"frames are {tuple(frames_u8.shape[1:3])}, the calibration is "
                f"{(self.cam_dist.image_height, self.cam_dist.image_width)}: the reference resizes with cv2.resize first "
                "(undistortion.py:171-173), which this build does not restate")
        mx, my = self.get_maps(target_size)
        return ops.remap_bilinear_u8(frames_u8.contiguous(), mx, my)


def create_undistortion_maps(cam_dist_path: str, device: str = "cuda") -> Optional[UndistortionMaps]:
    """utils/undistortion_utils.py:14-40 (scale is always 1.0 there)."""
    import os
    if not os.path.exists(cam_dist_path):
        print(f"❌ Camera calibration file not found: {cam_dist_path}")
        return None
    cam = Camera()
    cam.load_camera_calibration_file(cam_dist_path, 1.0)
    return UndistortionMaps(cam, device=device)
