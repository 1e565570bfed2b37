"""CPU tier: bench.py's own launcher.  `python bench.py --gpus N` with no WORLD_SIZE must start N ranks as a child
`torch.distributed.run` (the parent touches neither torch nor HIP); without a GPU every rank fails loudly - there is no CPU
fallback - and the launcher hands the failure back as its exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_starts_the_ranks_and_fails_loudly_without_gpu(ensure_built):
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present: the launcher is exercised by the gpu tier (test_distributed_gpu.py::test_bench_two_ranks_on_one_gpu)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "no HIP device visible" in out.stderr and "no CPU fallback" in out.stderr
    # both ranks were started by the child launcher
    assert out.stderr.count("bench.py needs an MI355X") >= 2 or "local_rank: 1" in out.stderr or "rank: 1" in out.stderr
    assert out.stdout.strip() == ""          # no JSON line from a run that measured nothing


def test_single_gpu_invocation_does_not_spawn(ensure_built):
    """--gpus 1 runs in-process (and fails loudly here: no device)."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no HIP device visible" in out.stderr and "torch.distributed.run" not in out.stderr


def test_pmc_evidence_is_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """The static roofline fields (traffic, dp_issue_util) come from committed rocprofv3 passes: bench.py only uses a record whose
    stored git blob hash of kde_kernels.hip equals the working tree's, and says why otherwise."""
    import json
    import subprocess as sp

    sys.path.insert(0, ROOT)
    import bench

    src = os.path.join(ROOT, "pybnesian_amd", "csrc", "kde_kernels.hip")
    blob = bench.git_blob_sha1(src)
    assert blob == sp.run(["git", "hash-object", src], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    fake = tmp_path / "repo"
    (fake / "profiles" / "r4").mkdir(parents=True)
    (fake / "pybnesian_amd" / "csrc").mkdir(parents=True)
    with open(src, "rb") as f:
        (fake / "pybnesian_amd" / "csrc" / "kde_kernels.hip").write_bytes(f.read())
    rec = {bench.SWEEP_KERNEL + ", true>": {"FETCH_SIZE": 100.0, "WRITE_SIZE": 10.0, "GRBM_GUI_ACTIVE": 8e6, "SQ_VALU_MFMA_BUSY_CYCLES": 1e8, "SQ_INSTS_VALU": 1e8},
           "_source_blob": {"kde_kernels.hip": blob}}
    (fake / "profiles" / "r4" / "pmc_per_dispatch.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(fake))
    k, why = bench.pmc_record(bench.SWEEP_KERNEL)
    assert why is None and k["FETCH_SIZE"] == 100.0
    assert bench.dp_issue_util()["value"] is not None
    with open(fake / "pybnesian_amd" / "csrc" / "kde_kernels.hip", "ab") as f:
        f.write(b"\n// edited\n")
    k, why = bench.pmc_record(bench.SWEEP_KERNEL)
    assert k is None and "stale" in why
    assert bench.dp_issue_util() == {"value": None, "reason": why}


def test_tie_accounting_classifies_flips():
    """bench.py's replay of a product trace inside the serial restatement: an exact tie taken the other way is a tie flip, a worse
    operator is a non-tie divergence, and where the trace stops early the restatement's remaining gain is reported."""
    sys.path.insert(0, ROOT)
    import bench

    class Op:
        def __init__(self, s, t, d):
            self._s, self._t, self._d = s, t, d

        def source(self):
            return self._s

        def target(self):
            return self._t

        def delta(self):
            return self._d

    class AddArc(Op):
        pass

    class RemoveArc(Op):
        pass

    class FlipArc(Op):
        pass

    class P:
        pass

    pbn = P()
    pbn.AddArc, pbn.RemoveArc, pbn.FlipArc = AddArc, RemoveArc, FlipArc
    names = ["a", "b", "c"]
    # decomposable toy score: a - b are score-equivalent (exact tie), c prefers parent a by less
    table = {(0, ()): 0.0, (1, ()): 0.0, (2, ()): 0.0, (0, (1,)): 2.0, (1, (0,)): 2.0, (2, (0,)): 1.0, (2, (1,)): 0.5, (0, (2,)): 1.0, (1, (2,)): 0.5}

    def sc(v, _t, par):
        return table.get((v, tuple(sorted(par))), -5.0)

    from oracle import hc_oracle

    own = [t[:3] for t in hc_oracle.estimate(3, 0, sc)[2]]
    first = own[0]
    flipped = (0, first[2], first[1])                      # the score-equivalent orientation of the first arc
    hc = P()
    hc.last = P()
    hc.last.trace = [AddArc(names[flipped[1]], names[flipped[2]], 2.0)]
    out = bench.tie_accounting(pbn, hc, names, sc)
    assert out["tie_flips"] == 1 and out["non_tie_divergences"] == 0 and out["end_gain"] > 0 and not out["product_graph_is_oracle_local_optimum"]
    hc.last.trace = [AddArc("b", "c", 0.5)]                # not the best operator and not a tie of it
    out = bench.tie_accounting(pbn, hc, names, sc)
    assert out["non_tie_divergences"] == 1 and out["max_gap"] == 1.5
    hc.last.trace = [AddArc(names[a], names[b], 0.0) for _, a, b in own]
    out = bench.tie_accounting(pbn, hc, names, sc)
    assert out["tie_flips"] == 0 and out["non_tie_divergences"] == 0 and out["product_graph_is_oracle_local_optimum"]


def test_compact_line_carries_every_leg_in_under_6000_bytes():
    """The driver's record keeps the parsed headline keys and the last ~6 KB of the stdout line: the line is numbers only and every leg's
    figures (C1, C3, C4 + Gram roofline, C5, f32, cv_weak, rccl_world1) are in it; prose lives in bench_full.json.  Input: the verbose
    object of a real run (profiles/r5/bench_default.json, 19 KB)."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "r5", "bench_default.json")) as fh:
        full = json.load(fh)
    line = json.dumps(bench.compact_line(full, "bench_full.json"), separators=(",", ":"))
    assert len(line) < 6000, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert out[k] == full[k] or abs(out[k] - full[k]) <= 1e-5 * abs(full[k])
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    legs = out["legs"]
    assert set(legs) >= {"c1", "c3", "c4", "c5", "f32", "cv_weak", "rccl_world1", "e2e_host"}
    assert legs["c3"]["estimate_s"] > 0 and legs["c5"]["estimate_s"] > 0 and legs["cv_weak"]["nodes"] == 23
    assert legs["c4"]["roofline"]["launch_us"] > 0 and legs["c4"]["ties"]["non_tie_divergences"] == 0
    assert legs["f32"]["roofline"]["frac"] > 0 and legs["f32"]["parity"]["ok"] is True
    assert legs["rccl_world1"]["ok"] is True and legs["rccl_world1"]["bit_identical"] is True

    def no_prose(o, path=""):
        for k, v in o.items():
            if isinstance(v, dict):
                no_prose(v, path + k + ".")
            elif isinstance(v, str):
                assert len(v) <= 140, (path + k, len(v))
    no_prose(out)
    assert out["full"] == "bench_full.json"
