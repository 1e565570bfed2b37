import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(0)
a = rng.normal(size=n)
df = pd.DataFrame({"a": a, "b": 0.5 * a + rng.normal(size=n), "c": rng.normal(size=n)})
test = pbn.KMutualInformation(df, 10, seed=0, samples=20)
for _ in range(20):
    test.mi("a", "b")
    test.mi("a", "b", "c")
