import json, os, subprocess, sys
import numpy as np
def run(env_extra):
    env = dict(os.environ); env.update(env_extra)
    p = subprocess.run([sys.executable, "tests/switch_worker_gpu.py"], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])
a = run({}); 
for name, env in (("lds1", {"PBN_GRAM_LDS": "1"}), ("nomirror", {"PBN_MI_MIRROR_MB": "0"}), ("nofull", {"PBN_MI_FULLGRAM": "0"})):
    b = run(env)
    for key in ("mi_plain", "mi_nulls"):
        x, y = np.array(a[key]), np.array(b[key])
        d = np.abs(x - y)
        i = int(np.argmax(d / np.maximum(np.abs(x), 1e-300)))
        print(name, key, "max abs", d.max(), "worst rel idx", i, x[i], y[i])
