"""Absolute pose error of a trajectory against ground truth: what the reference's evaluation scripts run,
`evo_ape tum <groundtruth> <estimate> -as` (scripts/eval_7scenes.sh:175, scripts/eval_euroc.sh), restated so the second
BASELINE metric (7-Scenes APE, README.md:73-85) can be produced where `evo` is not installed.

    python tools/eval_ape.py GROUNDTRUTH.txt ESTIMATE.txt [--max-diff 0.01] [--no-scale] [--no-align] [--json]

End to end, once the released weights and a 7-Scenes sequence are on the box:
    python -m pi3_slam_amd.cli create --images $SCENES/chess/seq-01/color/ --model-path $PI3_WEIGHTS --output out/chess \
        --chunk-length 100 --overlap 20 --metric-depth --keypoints grid --max-kp 400 --estimate-intrinsics --device-resize
    python -m pi3_slam_amd.cli reconstruct --chunks out/chess --output out/chess/reconstruction --max-observations-per-track 10
    python tools/eval_ape.py tests/golden/gt_7scenes_chess.txt out/chess/reconstruction/trajectory_tum.txt

What `evo_ape tum ref est -as` does (evo 1.x: main_ape.py, core/sync.py, core/trajectory.py, core/geometry.py, core/metrics.py):
  1. both files are TUM trajectories: `timestamp tx ty tz qx qy qz qw`, '#' comments;
  2. association by timestamp (sync.associate_trajectories, max_diff = 0.01 s, no offset): for every stamp of the SHORTER
     trajectory the closest stamp of the longer one, kept when |dt| <= max_diff;
  3. -a -s: Umeyama (1991) similarity of the estimate's positions onto the reference's, with scale
     (geometry.umeyama_alignment: covariance of the centred point sets, SVD, reflection fix, c = tr(D S) / sigma_x^2);
  4. pose relation `translation_part` (the default): e_i = |p_ref_i - p_est_aligned_i|; statistics rmse = sqrt(mean e^2),
     mean, median, std, min, max, sse.
This file restates exactly that (numpy only); it is evaluation tooling, not part of the hot path."""
from __future__ import annotations

import argparse
import json
import sys
from typing import Dict, Tuple

import numpy as np


def read_tum(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> stamps (n,), positions (n, 3), quaternions (n, 4) as (x, y, z, w).  Lines starting with '#' and blank lines are
    skipped; separators may be spaces or commas."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.replace(",", " ").split()
            if len(parts) != 8:
                raise ValueError(f"{path}: TUM trajectory files must have 8 entries per row (timestamp tx ty tz qx qy qz qw), "
                                 f"got {len(parts)}: {line[:80]!r}")
            rows.append([float(v) for v in parts])
    if not rows:
        raise ValueError(f"{path}: no poses")
    a = np.asarray(rows, np.float64)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def associate(stamps_ref: np.ndarray, stamps_est: np.ndarray, max_diff: float = 0.01) -> Tuple[np.ndarray, np.ndarray]:
    """Index pairs (into ref, into est): for every stamp of the shorter trajectory the closest stamp of the longer one,
    kept when the difference is <= max_diff (evo sync.matching_time_indices; several short stamps may pick the same long
    one; among equally close stamps the first in file order, as np.argmin does)."""
    est_longer = len(stamps_est) > len(stamps_ref)
    short, long_ = (stamps_ref, stamps_est) if est_longer else (stamps_est, stamps_ref)
    order = np.argsort(long_, kind="stable")
    ls = long_[order]
    pos = np.searchsorted(ls, short)
    i_short, i_long = [], []
    for i, (s, p) in enumerate(zip(short, pos)):
        cands = [c for c in (p - 1, p) if 0 <= c < len(ls)]
        best = min(cands, key=lambda c: (abs(ls[c] - s), order[c]))
        if abs(ls[best] - s) <= max_diff:
            i_short.append(i)
            i_long.append(order[best])
    i_short, i_long = np.asarray(i_short, int), np.asarray(i_long, int)
    if len(i_short) == 0:
        raise ValueError("found no matching timestamps between the reference and the estimate (max_diff "
                         f"{max_diff} s); 7-Scenes ground truth and this build's trajectory_tum.txt use frame indices as stamps")
    return (i_short, i_long) if est_longer else (i_long, i_short)


def umeyama(x: np.ndarray, y: np.ndarray, with_scale: bool = True) -> Tuple[np.ndarray, np.ndarray, float]:
    """Least-squares similarity y ~ c R x + t of point sets x, y (n, m) - Umeyama 1991, as evo geometry.umeyama_alignment."""
    if x.shape != y.shape or x.ndim != 2:
        raise ValueError("point sets must have the same (n, m) shape")
    n, m = x.shape
    mx, my = x.mean(0), y.mean(0)
    sigma_x = ((x - mx) ** 2).sum() / n
    cov = (y - my).T @ (x - mx) / n
    u, d, vt = np.linalg.svd(cov)
    if np.count_nonzero(d > np.finfo(d.dtype).eps) < m - 1:
        raise ValueError("degenerate covariance rank, Umeyama alignment is not possible")
    s = np.eye(m)
    if np.linalg.det(u) * np.linalg.det(vt) < 0.0:
        s[m - 1, m - 1] = -1.0
    r = u @ s @ vt
    c = float(np.trace(np.diag(d) @ s) / sigma_x) if with_scale else 1.0
    t = my - c * r @ mx
    return r, t, c


def ape(ref_path: str, est_path: str, max_diff: float = 0.01, align: bool = True, correct_scale: bool = True) -> Dict:
    """-> {'rmse', 'mean', 'median', 'std', 'min', 'max', 'sse', 'pairs', 'scale', 'rotation', 'translation'} (metres)."""
    s_ref, p_ref, _ = read_tum(ref_path)
    s_est, p_est, _ = read_tum(est_path)
    i_ref, i_est = associate(s_ref, s_est, max_diff)
    x, y = p_est[i_est], p_ref[i_ref]
    r, t, c = (np.eye(3), np.zeros(3), 1.0)
    if align:
        r, t, c = umeyama(x, y, correct_scale)
        x = c * x @ r.T + t
    e = np.linalg.norm(y - x, axis=1)
    return {"rmse": float(np.sqrt(np.mean(e ** 2))), "mean": float(e.mean()), "median": float(np.median(e)),
            "std": float(e.std()), "min": float(e.min()), "max": float(e.max()), "sse": float((e ** 2).sum()),
            "pairs": int(len(e)), "scale": c, "rotation": r.tolist(), "translation": t.tolist()}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="APE (translation part) with Sim(3) Umeyama alignment: evo_ape tum REF EST -as")
    ap.add_argument("ref")
    ap.add_argument("est")
    ap.add_argument("--max-diff", type=float, default=0.01, help="timestamp association threshold in seconds (evo: --t_max_diff)")
    ap.add_argument("--no-scale", action="store_true", help="SE(3) alignment (-a without -s)")
    ap.add_argument("--no-align", action="store_true", help="no alignment at all")
    ap.add_argument("--json", action="store_true", help="print the statistics as one JSON line")
    a = ap.parse_args(argv)
    try:
        res = ape(a.ref, a.est, a.max_diff, not a.no_align, not a.no_scale)
    except (ValueError, OSError) as e:
        print(f"eval_ape: {e}", file=sys.stderr)
        return 1
    if a.json:
        print(json.dumps(res))
    else:
        print(f"APE w.r.t. translation part (m), {'Sim(3)' if not a.no_scale else 'SE(3)'} Umeyama alignment, "
              f"{res['pairs']} pose pairs, scale correction {res['scale']:.6f}")
        for k in ("max", "mean", "median", "min", "rmse", "sse", "std"):
            print(f"{k:>10s}\t{res[k]:.6f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
