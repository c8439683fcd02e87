"""``python -m agent0.summary [root]`` — aggregate finished runs into CSV tables.

Counterpart of /root/reference agent0/summary.py:13-109: the reference globs ``*/*/best.pth`` + ``params.json`` below the current
directory (files that nothing at its HEAD writes any more) and produces ``summary.csv`` with the columns
exp_name, commit, algo, game, mean, std, max, min, size, frames (20-33), then per game a rank and a score table over
``<exp_name>_<algo>`` columns with an ``avg`` and a ``final`` rank row (62-107).  Here the per-run record is the checkpoint every run
leaves behind (``<logdir>/<run>/final.pth`` or ``best.pth``, Trainer.save_checkpoint: test returns "ITRs", "frame_count", "game", "algo",
"sha", "name"), the tables have the same columns, and the pandas dependency is replaced by the csv module.  Each run's
``progress.csv`` (one row per logged iteration, Trainer.logging) is the per-iteration twin of msg.log; ``progress_tail`` returns its
last row for dashboards.
"""
from __future__ import annotations

import csv
import glob
import os
import sys
from collections import OrderedDict
from typing import Dict, List

import numpy as np

SUMMARY_COLUMNS = ("exp_name", "commit", "algo", "game", "mean", "std", "max", "min", "size", "frames")
EXCLUDED_GAMES = ("Pong", "Asterix")      # summary.py:66: left out of the ranking


def read_runs(root: str) -> List[dict]:
    """One row per run directory below ``root`` (searched one and two levels deep, like the reference's ``*/*/best.pth``)."""
    import torch

    files = sorted(set(glob.glob(os.path.join(root, "*", "best.pth")) + glob.glob(os.path.join(root, "*", "final.pth")) +
                       glob.glob(os.path.join(root, "*", "*", "best.pth")) + glob.glob(os.path.join(root, "*", "*", "final.pth"))))
    rows, seen = [], set()
    for f in files:
        run_dir = os.path.dirname(f)
        if run_dir in seen and f.endswith("final.pth"):      # a best.pth of the same run wins, as in the reference
            continue
        seen.add(run_dir)
        blob = torch.load(f, map_location="cpu", weights_only=True)
        rs = np.asarray(blob.get("ITRs", []), dtype=np.float64)
        if rs.size == 0:
            continue
        rows.append(OrderedDict(exp_name=str(blob.get("name") or os.path.basename(os.path.dirname(run_dir))), commit=str(blob.get("sha", ""))[:6],
                                algo=str(blob.get("algo", "")), game=str(blob.get("game", "")), mean=float(rs.mean()), std=float(rs.std()), max=float(rs.max()),
                                min=float(rs.min()), size=int(rs.size), frames=int(blob.get("frame_count", 0))))
    return rows


def rank_tables(rows: List[dict]):
    """-> (rank rows, score rows) in the shape of summary.py:62-107: per game the runs sorted by mean (rank 0 = best), then the mean rank
    per column (``avg``) and the rank of that mean (``final``)."""
    games = []
    for r in rows:
        if r["game"] not in games and r["game"] not in EXCLUDED_GAMES:
            games.append(r["game"])
    ranks, scores = [], []
    for game in games:
        ordered = sorted((r for r in rows if r["game"] == game), key=lambda r: -r["mean"])
        rk, sc = {"game": game}, {"game": game}
        for i, r in enumerate(ordered):
            col = f"{r['exp_name']}_{r['algo']}"
            rk[col], sc[col] = i, r["mean"]
        ranks.append(rk)
        scores.append(sc)
    cols = sorted({c for rk in ranks for c in rk if c != "game"})
    mean_rank = {c: float(np.mean([rk[c] for rk in ranks if c in rk])) for c in cols}
    final = {c: i for i, c in enumerate(sorted(cols, key=lambda c: mean_rank[c]))}
    ranks.append(dict(mean_rank, game="avg"))
    ranks.append(dict(final, game="final"))
    return ranks, scores


def _write(path: str, rows: List[dict], columns=None):
    columns = list(columns) if columns else ["game"] + sorted({c for r in rows for c in r if c != "game"})
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([""] + columns)                      # pandas' to_csv writes the index as an unnamed first column
        for i, r in enumerate(rows):
            w.writerow([i] + ["" if r.get(c) is None else r.get(c) for c in columns])


def progress_tail(run_dir: str) -> Dict[str, str]:
    with open(os.path.join(run_dir, "progress.csv"), newline="") as f:
        rows = list(csv.DictReader(f))
    return rows[-1] if rows else {}


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    root = argv[0] if argv else os.getcwd()
    rows = read_runs(root)
    _write(os.path.join(root, "summary.csv"), rows, SUMMARY_COLUMNS)
    ranks, scores = rank_tables(rows)
    _write(os.path.join(root, "rank.csv"), ranks)
    _write(os.path.join(root, "score.csv"), scores)
    print(f"{len(rows)} runs -> summary.csv, rank.csv, score.csv in {root}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
