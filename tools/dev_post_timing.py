"""Where the non-forward milliseconds of a chunk step go (masks, MoGe, scale, intrinsics, keypoints + gather + D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.moge import MoGeEngine
from pi3_slam_amd.weights import Pi3Config
dev = "cuda:0"
eng = Pi3Engine(Pi3Config(), dev)
moge = MoGeEngine.from_pretrained("recipe", dev)
cc = OfflineCreatorConfig(model_path="recipe", output_dir="/tmp/pi3_post", chunk_length=100, overlap=20, device=dev,
                          do_metric_depth=True, keypoint_type="grid", max_num_keypoints=200, num_loader_workers=0)
cr = OfflineChunkCreator(cc, model=eng, moge_model=moge)
cr.target_size = (308, 406)
frames = torch.rand(1, 100, 3, 308, 406, device=dev)
paths = [[f"f{i}.png"] for i in range(100)]
for _ in range(2):
    cr._process_single_chunk(frames, paths)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); res = eng.forward(frames); t1 = T()
    masks = cr._compute_masks(res)[0]; t2 = T()
    md = moge.infer(frames[0, 0])["depth"]; t3 = T()
    sc = cr._get_scale_factor_for_pi3(md, res["local_points"][0, 0][..., 2], masks[0])
    ops.apply_scale(sc.reshape(1), res["local_points"], res["points"], res["camera_poses"]); t4 = T()
    cp = cr._estimate_camera_parameters(res); t5 = T()
    kp = cr.keypoint_extractor.extract(frames)
    dense = dict(points=res["points"][0], local_points=res["local_points"][0], conf=res["conf"][0], masks=masks, images=frames[0])
    interp = cr._interpolate_world_points_for_keypoints(dense, kp["keypoints"]); t6 = T()
    out = {k: v.cpu() for k, v in interp.items() if torch.is_tensor(v)}; poses = res["camera_poses"][0].cpu()
    cpc = {k: v.cpu() for k, v in cp.items()}; t7 = T()
    print(f"forward {1e3*(t1-t0):.1f} | masks {1e3*(t2-t1):.2f} | moge {1e3*(t3-t2):.2f} | scale {1e3*(t4-t3):.2f} | "
          f"intrinsics {1e3*(t5-t4):.2f} | keypoints+gather {1e3*(t6-t5):.2f} | D2H {1e3*(t7-t6):.2f} ms")
t0 = T(); cr._process_single_chunk(frames, paths); t1 = T()
print(f"_process_single_chunk total {1e3*(t1-t0):.1f} ms")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
cr._process_single_chunk(frames, paths); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
