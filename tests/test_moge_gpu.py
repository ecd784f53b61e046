"""GPU parity of the MoGe-2 metric-depth forward (C-ABI kernels) against the vectors produced by the REAL MoGeModel
class on the synthetic model_config + recipe weights (tests/golden/moge_*.npz).

Stated tolerance: the pipeline consumes only median(moge_depth / pi3_depth) over the valid mask
(slam/offline_chunk_creator.py:121-127), so the gate is on (a) the affine depth map: mean/max absolute error within 2x
the reference's own bf16-autocast deviation stored with the vectors (bf16err_z), (b) the binary mask: <= 0.5 % of the
pixels may flip (logits near 0), (c) the metric depth on the common mask: median relative error < 1 %, and (d) the
resulting median scale against a synthetic pi3 depth within 0.5 %.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(built_lib):
    assert torch.cuda.is_available()
    from pi3_slam_amd.moge import MoGeEngine
    return MoGeEngine.from_pretrained("recipe", "cuda:0")


@pytest.mark.parametrize("name", ["moge_small", "moge_chunk"])
def test_moge_infer_against_reference_vectors(engine, name):
    from oracle.gen_golden_moge import CASES, moge_image
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H, W, level = CASES[name]
    out = engine.infer(moge_image(name, H, W), resolution_level=level)
    torch.cuda.synchronize()
    z = out["points_affine"][..., 2].cpu().numpy()
    d = np.abs(z - g["points_affine_z"])
    assert d.mean() <= 2.0 * g["bf16err_z"][0] and d.max() <= 2.0 * g["bf16err_z"][1], (d.mean(), d.max(), g["bf16err_z"])
    mask_ref = np.unpackbits(g["mask"])[: H * W].reshape(H, W).astype(bool)
    mask = out["mask"].cpu().numpy()
    assert (mask != mask_ref).mean() < 5e-3
    both = mask & mask_ref
    depth = out["depth"].cpu().numpy()
    rel = np.abs(depth[both] - g["depth"][both]) / g["depth"][both]
    assert np.median(rel) < 1e-2 and rel.max() < 8e-2, (np.median(rel), rel.max())
    assert np.all(np.isinf(depth[~mask])) and np.all(np.isfinite(depth[mask]))
    # what the pipeline does with it: the median ratio against a pi3 depth map
    pi3_z = (g["depth"] / 1.37).astype(np.float32)
    pi3_z[~np.isfinite(pi3_z)] = 1.0
    s_ref = np.median((g["depth"] / pi3_z)[both])
    s = np.median((depth / pi3_z)[both])
    assert abs(s - s_ref) / s_ref < 5e-3
    np.testing.assert_allclose(out["intrinsics"].cpu().numpy(), g["intrinsics"], rtol=2e-2)


def test_moge_rejects_unbuilt_config():
    from pi3_slam_amd.moge import SYNTHETIC_CONFIG, moge_param_shapes
    import copy
    cfg = copy.deepcopy(SYNTHETIC_CONFIG)
    cfg["neck"]["resamplers"] = ["pixel_shuffle"] * 4
    with pytest.raises(NotImplementedError):
        moge_param_shapes(cfg)
