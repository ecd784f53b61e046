"""Generates tests/golden/ingest_*.npz by running Pillow (the third-party code behind the reference's
transforms.Resize, datasets/image_datasets.py:186-208) on seeded random frames: inputs, the resized uint8 frames and
the ToTensor output.  Run from the repo root:  python oracle/gen_golden_ingest.py"""
import os

import numpy as np
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"ingest_down": (2, 96, 150, 56, 84), "ingest_up": (1, 40, 52, 70, 98), "ingest_mixed": (2, 64, 30, 28, 42)}


def main():
    import PIL
    rng = np.random.default_rng(2024)
    for name, (N, H0, W0, H1, W1) in CASES.items():
        frames = rng.integers(0, 256, (N, H0, W0, 3), dtype=np.uint8)
        frames[0, : H0 // 2] //= 4                         # a dark half so that clipping / rounding at 0 is hit
        out = np.stack([np.array(Image.fromarray(f).resize((W1, H1), Image.BILINEAR)) for f in frames])
        tens = (out.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)
        np.savez_compressed(os.path.join(REPO, "tests", "golden", name + ".npz"), frames=frames, resized=out,
                            tensor=tens, target=np.array([H1, W1]), pillow=np.array(PIL.__version__))
        print(name, frames.shape, "->", out.shape)


if __name__ == "__main__":
    main()
