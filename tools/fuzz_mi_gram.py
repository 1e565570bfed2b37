"""Shape fuzz of the gathered Gram of MutualInformation groupings (gram_gring_kernel through the row-major mirror / through the columns)
against the per-test moment kernels: random rows (1 ... 70 000), continuous columns (1 ... 64), dtypes, cardinalities and configuration
sizes (empty and one-dominant configurations included; no configuration of 1-39 rows: singular covariances are noise on every path).  Every case runs in three processes - default (mirror),
PBN_MI_MIRROR_MB=0 (columns), PBN_MI_FULLGRAM=0 (per-test kernels, the reference here) - on the same seeded table and tests.
    python3 tools/fuzz_mi_gram.py [cases, default 24] [seed]"""
import json, os, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else %r)
import numpy as np, pandas as pd
import pybnesian_amd as pbn
cases = json.loads(sys.argv[1])
out = []
for cs in cases:
    rng = np.random.default_rng(cs["seed"])
    n, d, dtype = cs["rows"], cs["cols"], cs["dtype"]
    mix = np.eye(d) + 0.2 * np.tril(rng.normal(size=(d, d)), -1)
    x = (rng.normal(size=(n, d)) @ mix.T * rng.uniform(0.5, 2.0, size=d) + rng.uniform(-20, 20, size=d)).astype(dtype)
    df = pd.DataFrame(x, columns=[f"x{i}" for i in range(d)])
    # joint configurations with uneven shares; a configuration is EMPTY or holds at least 40 rows (a handful of rows under several
    # continuous variables is a singular covariance: every path returns noise there, each its own)
    cards = cs["cards"]
    nj = int(np.prod(cards))
    joint = rng.choice(nj, size=n, p=rng.dirichlet(np.full(nj, cs["alpha"])))
    cnt = np.bincount(joint, minlength=nj)
    big = int(np.argmax(cnt))
    joint[np.isin(joint, np.where((cnt > 0) & (cnt < 40))[0])] = big
    codes, rest = [], joint.copy()
    for j, card in enumerate(cards):
        c = rest %% card
        rest = rest // card
        codes.append(c)
        df[f"d{j}"] = pd.Categorical.from_codes(c, [f"k{i}" for i in range(card)])
    if d > 1:
        df["x0"] = (df["x0"] + 0.5 * codes[0]).astype(dtype)
    t = pbn.MutualInformation(df)
    names = list(df.columns)
    r2 = np.random.default_rng(cs["seed"] + 1)
    vals = []
    for _ in range(cs["tests"]):
        k = int(r2.integers(1, 4))
        sel = [names[i] for i in r2.choice(len(names), size=min(k + 2, len(names)), replace=False)]
        if not any(s.startswith("d") for s in sel[2:]):
            sel[-1] = "d0" if "d0" not in sel[:2] else ("d1" if len(cs["cards"]) > 1 and "d1" not in sel[:2] else sel[-1])
        if len(set(sel)) < len(sel) or len(sel) < 3:
            continue
        try:
            vals.append(t.mi(sel[0], sel[1], sel[2:]))
        except Exception as ex:
            vals.append("ERR " + type(ex).__name__)
    out.append(vals)
print("RESULT " + json.dumps(out))
''' % os.path.dirname(HERE)


def run(cases, env_extra):
    env = dict(os.environ); env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", WORKER, json.dumps(cases)], env=env, capture_output=True, text=True, timeout=1500)
    if p.returncode != 0:
        print(p.stderr[-3000:]); raise SystemExit(1)
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
cases = []
for i in range(ncases):
    rows = int(rng.choice([int(rng.integers(60, 400)), int(rng.integers(400, 9000)), int(rng.integers(9000, 70000))]))
    ncard = int(rng.integers(1, 3))
    cases.append({"seed": 100 + i, "rows": rows, "cols": int(rng.choice([1, 2, 5, 16, 17, 31, 33, 48, 49, 64, int(rng.integers(1, 65))])),
                  "dtype": "float64" if rng.random() < 0.6 else "float32", "cards": [int(rng.integers(2, 9)) for _ in range(ncard)],
                  "alpha": float(rng.choice([0.05, 0.5, 5.0])), "tests": 10})
ref = run(cases, {"PBN_MI_FULLGRAM": "0"})
worst = {"mirror": 0.0, "columns": 0.0, "lds-image": 0.0}
bad = 0
for tag, env in (("mirror", {}), ("columns", {"PBN_MI_MIRROR_MB": "0"}), ("lds-image", {"PBN_GRAM_LDS": "1"})):
    got = run(cases, env)
    for cs, a, b in zip(cases, got, ref):
        tol = 1e-8 if cs["dtype"] == "float64" else 2e-3
        for u, v in zip(a, b):
            if isinstance(u, str) or isinstance(v, str):
                if u != v:
                    bad += 1; print("MISMATCH (error)", tag, cs, u, v)
                continue
            if not (np.isfinite(u) and np.isfinite(v)):
                if not ((np.isnan(u) and np.isnan(v)) or u == v):
                    bad += 1; print("MISMATCH (non-finite)", tag, cs, u, v)
                continue
            err = abs(u - v) / max(abs(v), 1e-6)
            worst[tag] = max(worst[tag], err if cs["dtype"] == "float64" else 0.0)
            if err > tol:
                bad += 1; print("MISMATCH", tag, cs, u, v, err)
print(f"{ncases} shapes x 10 tests x 3 paths against the per-test kernels: {'all ok' if not bad else str(bad) + ' MISMATCHES'}; worst fp64 relative difference "
      f"mirror {worst['mirror']:.2e}, columns {worst['columns']:.2e}, LDS-image kernel {worst['lds-image']:.2e}")
sys.exit(1 if bad else 0)
