"""ORACLE tooling — generates tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Run in the build container only (the reference does not exist on the GPU box):
    python oracle/gen_golden.py            # pi3 network vectors (needs ~10 GB RAM, ~2 min)
The reference's pi3 classes are imported unmodified; weights are the recipe weights (pi3_slam_amd/recipe.py) loaded
through load_state_dict, inputs are recipe streams too, so tests regenerate both and the .npz only carries outputs
and a few intermediate activations (captured with forward hooks).  The script also runs the oracle restatement
(oracle/pi3_ref.py) on the same data and prints the differences, which is the "pin" of the oracle.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from pi3_slam_amd.recipe import recipe_unit, fnv1a64  # noqa: E402
from pi3_slam_amd.weights import Pi3Config, recipe_state_dict_cpu  # noqa: E402

CASES = {
    # name: (B, N, H, W)
    "pi3_tiny_a": (1, 3, 28, 42),
    "pi3_tiny_b": (1, 2, 42, 56),
    "pi3_tiny_c": (1, 5, 56, 70),
    # 8 frames at the size every 4:3 input becomes (308 x 406): S = 5 144 tokens, M = 5 144 rows.  At this size the HIP
    # path runs the kernels that ship (256x256 GEMM, 64-row attention with four-wave workgroups on the 643-token frame
    # sequences and eight-wave workgroups on the global one); the tiny cases above go through the small-shape kernels.
    # Dense maps are stored every SUB-th pixel, intermediates every ROWS-th token (the file stays < 4 MB).
    "pi3_mid": (1, 8, 308, 406),
    # the same chunk shape with the decoder's q / k LayerNorm gains x HOT_GAIN (scores x 12): |q| max|k| leaves the
    # a-priori bound of the HIP attention's loop without a running maximum (attn64.hip) in every decoder block, frame-wise
    # and global, the way real weights might - the reference's answer to what that loop must still reproduce
    "pi3_mid_hot": (1, 8, 308, 406),
}
SUBSAMPLED = {"pi3_mid": (7, 64), "pi3_mid_hot": (7, 64)}     # name: (pixel stride SUB, token-row stride ROWS)
HOT_GAIN = 3.5


def hot_overrides(sd: dict, name: str) -> dict:
    """Weight edits of a case (tensor name -> new tensor); empty for the plain recipe cases.  The GPU test applies the
    same edit to the engine's weights."""
    if name != "pi3_mid_hot":
        return {}
    return {k: v * HOT_GAIN for k, v in sd.items()
            if k.startswith("decoder.") and k.endswith(("q_norm.weight", "k_norm.weight"))}


def golden_images(name: str, B: int, N: int, H: int, W: int) -> torch.Tensor:
    """Deterministic test frames in [0, 1): smooth-ish (low-frequency ramp + recipe noise) so patches differ."""
    u = recipe_unit(fnv1a64("golden.images." + name), B * N * 3 * H * W).reshape(B, N, 3, H, W)
    yy = np.linspace(0, 1, H, dtype=np.float32)[None, None, None, :, None]
    xx = np.linspace(0, 1, W, dtype=np.float32)[None, None, None, None, :]
    img = 0.5 + 0.25 * (u.astype(np.float32)) + 0.2 * (yy - 0.5) + 0.1 * (xx - 0.5)
    return torch.from_numpy(np.clip(img, 0.0, 1.0).astype(np.float32))


def main() -> None:
    sys.path.insert(0, REF)
    from pi3.models.pi3 import Pi3  # the real reference network
    from oracle import pi3_ref

    cfg = Pi3Config()
    t0 = time.time()
    sd = recipe_state_dict_cpu(cfg)
    print(f"recipe weights: {len(sd)} tensors in {time.time() - t0:.1f}s")
    model = Pi3().eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"image_mean", "image_std"}, (missing, unexpected)

    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[1:]
    for name, (B, N, H, W) in CASES.items():
        if only and name not in only:
            continue
        sub, rows = SUBSAMPLED.get(name, (1, 1))
        imgs = golden_images(name, B, N, H, W)
        plain_sd = sd
        edits = hot_overrides(plain_sd, name)
        if edits:
            sd = dict(plain_sd, **edits)
            model.load_state_dict(sd, strict=False)
            print(f"{name}: {len(edits)} tensors edited")
        cap = {}
        hooks = [
            model.encoder.blocks[0].register_forward_pre_hook(lambda m, a: cap.__setitem__("tokens", a[0].detach().clone())),
            model.decoder[0].register_forward_pre_hook(lambda m, a: cap.__setitem__("enc_out", a[0].detach().clone())),
            model.decoder[0].register_forward_hook(lambda m, a, o: cap.__setitem__("dec0", o.detach().clone())),
            model.decoder[1].register_forward_hook(lambda m, a, o: cap.__setitem__("dec1", o.detach().clone())),
            model.point_decoder.register_forward_pre_hook(lambda m, a: cap.__setitem__("dec_cat", a[0].detach().clone())),
            model.point_decoder.register_forward_hook(lambda m, a, o: cap.__setitem__("point_decoder", o.detach().clone())),
            model.conf_decoder.register_forward_hook(lambda m, a, o: cap.__setitem__("conf_decoder", o.detach().clone())),
            model.camera_decoder.register_forward_hook(lambda m, a, o: cap.__setitem__("camera_decoder", o.detach().clone())),
        ]
        t0 = time.time()
        with torch.no_grad():
            ref = model(imgs)
        for h in hooks:
            h.remove()
        print(f"{name}: reference forward {time.time() - t0:.1f}s")
        t0 = time.time()
        orc = pi3_ref.pi3_forward(sd, imgs, cfg, return_intermediates=True)
        print(f"{name}: oracle forward {time.time() - t0:.1f}s")
        save = {"shape": np.array([B, N, H, W]), "strides": np.array([sub, rows])}
        for k in ("points", "local_points", "conf", "camera_poses"):
            save[k] = ref[k].numpy() if k == "camera_poses" else ref[k][:, :, ::sub, ::sub].numpy()
            d = (ref[k] - orc[k]).abs().max().item()
            print(f"   {k:14s} ref-vs-oracle max|d| = {d:.3e}   (ref max {ref[k].abs().max().item():.3f})")
        for k, v in cap.items():
            v2 = v.reshape(-1, v.shape[-1])
            save["i_" + k] = v2[::rows].numpy()
            d = (v2 - orc["_intermediates"][k]).abs().max().item()
            print(f"   i_{k:12s} ref-vs-oracle max|d| = {d:.3e}   (ref max {v2.abs().max().item():.3f})")
        # Tolerance anchor: the reference's OWN bf16-autocast execution vs its fp32 execution on the same weights
        # (what offline_chunk_creator.py:168-171 runs on a GPU).  The reference disables autocast for the heads with
        # device_type='cuda' contexts (pi3.py:192, camera_head.py:68); map those to the CPU autocast so they are honoured.
        orig_autocast = torch.amp.autocast

        class _CpuAutocast(orig_autocast):
            def __init__(self, device_type, *a, **k):
                super().__init__("cpu" if device_type == "cuda" else device_type, *a, **k)

        torch.amp.autocast = _CpuAutocast
        try:
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                ref16 = model(imgs)
        finally:
            torch.amp.autocast = orig_autocast
        for k in ("points", "local_points", "conf", "camera_poses"):
            d = (ref16[k].float() - ref[k]).abs()
            save["bf16err_" + k] = np.array([d.mean().item(), d.max().item()], dtype=np.float64)
            print(f"   {k:14s} reference bf16-autocast vs fp32: mean|d| {d.mean().item():.3e} max|d| {d.max().item():.3e}"
                  f"  (dtype {ref16[k].dtype})")
        Ra, Rb = ref16["camera_poses"][0, :, :3, :3].double(), ref["camera_poses"][0, :, :3, :3].double()
        tr = ((Ra @ Rb.transpose(-1, -2)).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
        save["bf16err_rot_deg"] = np.array([torch.rad2deg(torch.acos(tr.clamp(-1, 1))).max().item()])
        print("   reference bf16 rotation error (deg):", save["bf16err_rot_deg"])
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **save)
        print("   wrote", os.path.join(out_dir, name + ".npz"))
        if edits:
            sd = plain_sd
            model.load_state_dict(sd, strict=False)


if __name__ == "__main__":
    main()
