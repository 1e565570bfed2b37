"""Tie-flip report (SURVEY.md §7 hard part a): BIC / BGe hill-climbs without an orientation blacklist on the reference's
4-variable table and on 16- / 32-node synthetic Gaussian tables; for every case the product's operator trace is replayed
in the oracle restatement and every step where the oracle's own greedy choice differs is listed with the gap between the two
deltas (a tie when <= 1e-9 relative).  python tools/tie_flips.py > profiles/r2/tie_flips.json   (GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pybnesian_amd as pbn  # noqa: E402
import test_tieflip_gpu as tf  # noqa: E402
from helpers import frame  # noqa: E402

golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_recipes.npz"))
cases = [("reference 4-variable table, 2000 rows", frame(golden["train10k"][:2000])),
         ("16-node linear-Gaussian DAG, 5000 rows", tf.dag_table(5000, 16, 21)),
         ("32-node linear-Gaussian DAG, 20000 rows", tf.dag_table(20000, 32, 22))]
out = []
for label, df in cases:
    for kind in ("bic", "bge"):
        r = tf.run_case(pbn, df, kind)
        r["table"] = label
        r["n_flips"] = len(r["flips"])
        out.append(r)
print(json.dumps(out, indent=1))
