"""masks_kernel: the fast-decision form against the exact-only form (PI3_MASKS_EXACT_ONLY=1), same process, same data:
equality of every decision on adversarial inputs, and the time of each at the headline size (100 x 308 x 406)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd import ops  # noqa: E402


def both(conf, lp, thr, rtol):
    os.environ.pop("PI3_MASKS_EXACT_ONLY", None)
    a = ops.compute_masks(conf, lp, thr, rtol)
    os.environ["PI3_MASKS_EXACT_ONLY"] = "1"
    b = ops.compute_masks(conf, lp, thr, rtol)
    os.environ.pop("PI3_MASKS_EXACT_ONLY", None)
    return a, b


def timeit(conf, lp, n=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ops.compute_masks(conf, lp, 0.1, 0.03)
    ev[0].record()
    for _ in range(n):
        ops.compute_masks(conf, lp, 0.1, 0.03)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


def main():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    bad = 0
    for (F, H, W) in [(100, 308, 406), (100, 280, 448), (3, 17, 5), (2, 14, 342), (1, 1, 1), (2, 29, 1024)]:
        for case in range(6):
            lp = torch.randn(F, H, W, 3, device=dev, generator=g)
            z = torch.exp(0.3 * torch.randn(F, H, W, device=dev, generator=g))
            if case == 1:       # smooth depth: ratios near the threshold
                yy = torch.arange(H, device=dev).view(1, H, 1) * 0.0148
                xx = torch.arange(W, device=dev).view(1, 1, W) * 0.0151
                z = 1.0 + yy + xx + 1e-4 * torch.randn(F, H, W, device=dev, generator=g)
            if case == 2:       # specials sprinkled in
                r = torch.rand(F, H, W, device=dev, generator=g)
                z = torch.where(r < 0.01, torch.full_like(z, float("nan")), z)
                z = torch.where((r > 0.01) & (r < 0.02), torch.full_like(z, float("inf")), z)
                z = torch.where((r > 0.02) & (r < 0.03), torch.zeros_like(z), z)
                z = torch.where((r > 0.03) & (r < 0.04), -z, z)
                z = torch.where((r > 0.04) & (r < 0.05), z * 1e-42, z)
                z = torch.where((r > 0.05) & (r < 0.06), z * 1e35, z)
            if case == 3:       # inf / zero / tiny but no NaN (the fast form stays on)
                r = torch.rand(F, H, W, device=dev, generator=g)
                z = torch.where(r < 0.01, torch.full_like(z, float("inf")), z)
                z = torch.where((r > 0.02) & (r < 0.03), torch.zeros_like(z), z)
                z = torch.where((r > 0.03) & (r < 0.04), -z, z)
                z = torch.where((r > 0.04) & (r < 0.05), z * 1e-42, z)
                z = torch.where((r > 0.05) & (r < 0.06), z * 1e35, z)
                z = torch.where((r > 0.06) & (r < 0.07), torch.full_like(z, float("-inf")), z)
            lp[..., 2] = z
            conf = torch.randn(F, H, W, 1, device=dev, generator=g) * 3
            thr, rtol = 0.1, 0.03
            if case == 4:       # confidences packed around the crossing of the sigmoid
                c0 = -torch.log(torch.tensor(9.0)).item()
                k = torch.randint(-200, 200, (F, H, W, 1), device=dev, generator=g, dtype=torch.int32)
                conf = (torch.full((F, H, W, 1), c0, device=dev).view(torch.int32) + k).view(torch.float32)
            if case == 5:
                thr, rtol = [(0.5, 0.0), (0.9, 1.0), (-1.0, 0.03), (1.0, 0.03), (1e-30, 3e38), (0.1, float("inf"))][(F + H) % 6]
                conf[0, 0, 0, 0] = float("nan")
                conf[-1, -1, -1, 0] = float("inf")
            a, b = both(conf.contiguous(), lp.contiguous(), thr, rtol)
            n = (a != b).sum().item()
            bad += n
            print(f"{F}x{H}x{W} case {case} thr {thr} rtol {rtol}: differing decisions {n}, kept {a.float().mean().item():.3f}")
    F, H, W = 100, 308, 406
    lp = torch.randn(F, H, W, 3, device=dev, generator=g)
    lp[..., 2] = 1.0 + torch.arange(W, device=dev).view(1, 1, W) * 0.01 + 0.02 * torch.rand(F, H, W, device=dev, generator=g)
    conf = torch.randn(F, H, W, 1, device=dev, generator=g) * 3
    t_fast = timeit(conf, lp)
    os.environ["PI3_MASKS_EXACT_ONLY"] = "1"
    t_exact = timeit(conf, lp)
    os.environ.pop("PI3_MASKS_EXACT_ONLY", None)
    nbytes = F * H * W * 17
    print(f"masks 100x308x406: fast {t_fast:.1f} us ({nbytes / t_fast / 1e6:.2f} TB/s)  exact-only {t_exact:.1f} us "
          f"({nbytes / t_exact / 1e6:.2f} TB/s)")
    print("DIFF TOTAL", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
