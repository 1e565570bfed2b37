"""Diagnostic: does torch still see the GPU after the library has used it in the same process (and after a child process has)?"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import pybnesian_amd as pbn
df = pd.DataFrame(np.random.default_rng(0).normal(size=(1000, 3)), columns=list("abc"))
k = pbn.KDE(list("abc")); k.fit(df); print("slogl", k.slogl(df))
subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); import pybnesian_amd as p; p.load_library(); print('child ok')"])
import torch
print("torch.cuda.is_available after library use:", torch.cuda.is_available(), torch.cuda.device_count())
