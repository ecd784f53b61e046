"""ORACLE tooling — generates tests/golden/chunk_mid.npz by running the REAL reference's per-chunk function,
`OfflineChunkCreator._process_single_chunk` (slam/offline_chunk_creator.py:161-256), imported from /root/reference
(build container only), on 8 frames at 308x406 with the recipe weights: the chunk dictionary a user of the reference
gets - keys, dtypes, shapes, values - for the same inputs the HIP path's `_process_single_chunk` is given in
tests/test_pipeline_gpu.py.

    python oracle/gen_golden_chunk.py        (~3 minutes, ~10 GB RAM)

The object is built with object.__new__ (its __init__ fetches checkpoints by name); the model is the reference's Pi3
class with the recipe state dict; grid keypoints with max_num_keypoints = 4096, for which the reference's grid (spacing
clamped to 8 px: 35 x 46 = 1610 points) needs no random subset, so the run is deterministic; MoGe is absent (it is
hard-wired to 'cuda' in the reference, quirk 6 of SURVEY §8).  Absent third-party modules get empty placeholders as in
gen_golden_post.py; `utils3d.torch.intrinsics_from_focal_center` is restated from its call site
(utils/camera_estimation.py:56-57) so that the reference's intrinsics estimation runs instead of raising."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle.gen_golden import golden_images  # noqa: E402
from oracle.gen_golden_post import _Placeholder  # noqa: E402
from pi3_slam_amd.weights import Pi3Config, recipe_state_dict_cpu  # noqa: E402

CASE = ("chunk_mid", 8, 308, 406, 4096)      # name, frames, H, W, max_num_keypoints


def main() -> None:
    name, N, H, W, max_kp = CASE
    for mod in ("cv2", "natsort", "plyfile", "torchvision", "torchvision.transforms", "torchcodec",
                "torchcodec.decoders", "pytheia"):
        if mod not in sys.modules:
            sys.modules[mod] = _Placeholder(mod)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    u3, u3t = types.ModuleType("utils3d"), types.ModuleType("utils3d.torch")

    def intrinsics_from_focal_center(fx, fy, cx, cy):
        K = torch.zeros(*fx.shape, 3, 3, dtype=fx.dtype)
        K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2], K[..., 2, 2] = fx, fy, cx, cy, 1.0
        return K

    u3t.intrinsics_from_focal_center = intrinsics_from_focal_center
    u3.torch = u3t
    sys.modules["utils3d"], sys.modules["utils3d.torch"] = u3, u3t
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    sys.path.insert(0, REF)
    os.chdir(REF)
    from pi3.models.pi3 import Pi3
    from slam.offline_chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
    from utils.keypoint_extraction import create_keypoint_extractor

    model = Pi3().eval()
    missing, unexpected = model.load_state_dict(recipe_state_dict_cpu(Pi3Config()), strict=False)
    assert not unexpected and set(missing) <= {"image_mean", "image_std"}
    cr = object.__new__(OfflineChunkCreator)
    cr.config = OfflineCreatorConfig(model_path="unused", output_dir="/tmp/pi3_chunk_golden", chunk_length=N, overlap=2,
                                     device="cpu", do_metric_depth=False, keypoint_type="grid",
                                     max_num_keypoints=max_kp, estimate_camera_params=True)
    cr.model, cr.moge_model, cr.undistortion_maps = model, None, None
    cr.keypoint_extractor = create_keypoint_extractor(keypoint_type="grid", max_num_keypoints=max_kp,
                                                      detection_threshold=0.005, device="cpu")
    cr.target_size = (H, W)
    imgs = golden_images("pi3_mid", 1, N, H, W)          # the frames of tests/golden/pi3_mid.npz
    res = cr._process_single_chunk(imgs, [[f"frame_{i:03d}.png"] for i in range(N)])
    save = {"shape": np.array([N, H, W, max_kp])}
    schema = []
    for k, v in res.items():
        if torch.is_tensor(v):
            schema.append(f"{k}:{str(v.dtype).replace('torch.', '')}:{'x'.join(str(d) for d in v.shape)}")
            save[k] = v.view(torch.int16).numpy() if v.dtype == torch.float16 else v.numpy()
        elif isinstance(v, dict) and k == "camera_params":
            for kk, vv in v.items():
                schema.append(f"camera_params.{kk}:{str(vv.dtype).replace('torch.', '')}:{'x'.join(str(d) for d in vv.shape)}")
                save["camera_params." + kk] = vv.numpy()
        else:
            schema.append(f"{k}:{type(v).__name__}")
    save["schema"] = np.array(sorted(schema))
    print("\n".join(sorted(schema)))
    print("mask fraction at keypoints", float(res["masks"].float().mean()), "K", res["keypoints"].shape[1])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", name + ".npz"), **save)
    print("wrote", name)


if __name__ == "__main__":
    main()
