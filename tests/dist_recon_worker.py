"""Worker of tests/test_pipeline_gpu.py::test_reconstruct_with_bundle_adjust_under_torchrun_is_bounded: stage 2 only
(OfflineReconstructor with bundle_adjust=True) over an existing chunk directory; single process or under
torch.distributed.run with PI3_DIST_BACKEND=gloo."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd.reconstructor import OfflineReconstructor  # noqa: E402


def main():
    chunk_dir, out_dir = sys.argv[1:3]
    rec = OfflineReconstructor(chunk_dir, out_dir, bundle_adjust=True)
    rec.run()
    if int(os.environ.get("RANK", "0")) == 0:
        json.dump({"stages": rec.refinement_stages, "ba": [bool(i.get("success")) for i in rec.ba_infos]},
                  open(os.path.join(out_dir, "stages.json"), "w"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
