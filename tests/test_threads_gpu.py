"""GPU tier: locking of the C ABI (csrc/common.hpp: one recursive mutex per context; a search holds only its own state's lock).
ctypes drops the GIL around library calls, so Python threads really are inside the library at the same time."""
import threading

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def _tables(seed, n=40000, m=6000):
    rng = np.random.default_rng(seed)
    mix = np.array([[1, 0, 0], [0.5, 1, 0], [0.2, -0.4, 1.0]])
    return (pd.DataFrame(rng.normal(size=(n, 3)) @ mix.T, columns=list("abc")), pd.DataFrame(rng.normal(size=(m, 3)) @ mix.T, columns=list("abc")))


def test_threads_on_two_contexts_and_on_one(pbn):
    """Four threads, two contexts on the one device: threads 0 / 1 share context A, threads 2 / 3 context B; every thread fits and
    evaluates its own KDE repeatedly.  Same-context calls are serialised (they share scratch arenas), different contexts run side by
    side; every result must equal the one a single thread computes."""
    ctxs = [pbn.Context(0), pbn.Context(0)]
    want, out, errors = {}, {}, []
    for i in range(4):
        tr, te = _tables(100 + i)
        k = pbn.KDE(list("abc"))
        k.fit(tr)
        want[i] = (k.slogl(te), np.asarray(k.logl(te.iloc[:64])))

    def work(i):
        try:
            ctx = ctxs[i // 2]
            tr, te = _tables(100 + i)
            ttab, _ = pbn.DeviceTable.from_dataframe(ctx, tr, list("abc"))
            qtab, _ = pbn.DeviceTable.from_dataframe(ctx, te, list("abc"))
            vals = []
            for _ in range(12):
                k = pbn.KDE(list("abc"))
                k.fit_table(ttab)
                vals.append(k.slogl_table(qtab))
            out[i] = vals
        except Exception as ex:  # surfaced below
            errors.append((i, ex))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for i in range(4):
        assert len(out[i]) == 12 and all(v == out[i][0] for v in out[i])          # bit-identical run to run
        assert abs(out[i][0] - want[i][0]) <= 1e-12 * abs(want[i][0])


def test_another_thread_gets_through_during_a_search(pbn):
    """A hill-climb no longer holds a process-wide lock for its whole run: while one thread searches with a (slow) Python score,
    another thread's library calls on the default context complete long before the search does."""
    import time

    rng = np.random.default_rng(5)
    x = rng.normal(size=(3000, 5))
    x[:, 1] += x[:, 0]
    x[:, 3] -= 0.7 * x[:, 2]
    df = pd.DataFrame(x, columns=list("abcde"))
    bic = pbn.BIC(df)

    class Slow(pbn.Score):
        def has_variables(self, v):
            return True

        def compatible_bn(self, m):
            return True

        def local_score(self, model, variable, evidence=None):
            time.sleep(0.02)
            return bic.local_score(model, variable, model.parents(variable) if evidence is None else evidence)

    done = {}

    def search():
        res = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), Slow(), pbn.GaussianNetwork(list("abcde")))
        done["search"] = (time.perf_counter(), res.num_arcs())

    t = threading.Thread(target=search)
    t0 = time.perf_counter()
    t.start()
    time.sleep(0.05)
    tr, te = _tables(9, 20000, 2000)
    k = pbn.KDE(list("abc"))
    k.fit(tr)
    val = k.slogl(te)
    t_other = time.perf_counter()
    t.join(timeout=300)
    assert np.isfinite(val) and done["search"][1] >= 2
    assert t_other < done["search"][0], "the other thread's calls waited for the whole search"
    assert done["search"][0] - t0 > 0.3   # the search was long enough for the comparison to mean something


def test_handles_may_outlive_their_context(pbn):
    """A garbage collector finalises the objects of a reference cycle in no particular order: the context's destroy call may arrive
    before those of the tables and models created on it.  Every handle holds a counted reference (csrc/common.hpp, "Lifetime"), so
    the late destroy calls find a live mutex - and the handles keep working until then."""
    import ctypes as C
    import gc

    from pybnesian_amd import _lib

    lib = _lib.load()
    for _ in range(20):
        ctx = pbn.Context(0)
        tr, te = _tables(3, 5000, 500)
        ttab, _ = pbn.DeviceTable.from_dataframe(ctx, tr, list("abc"))
        qtab, _ = pbn.DeviceTable.from_dataframe(ctx, te, list("abc"))
        k = pbn.KDE(list("abc"))
        k.fit_table(ttab)
        want = k.slogl_table(qtab)
        handle, ctx.handle = ctx.handle, None          # the context's destroy call comes first ...
        lib.pbn_ctx_destroy(handle)
        assert k.slogl_table(qtab) == want             # ... the handles still work ...
        del k, qtab, ttab                              # ... and their destroy calls release the context with the last of them
        gc.collect()

    class Node:   # the same through a real cycle: context, table and model only reachable from it
        pass

    for _ in range(20):
        n = Node()
        n.self = n
        n.ctx = pbn.Context(0)
        n.tab, _ = pbn.DeviceTable.from_dataframe(n.ctx, _tables(4, 3000, 10)[0], list("abc"))
        n.k = pbn.KDE(list("abc"))
        n.k.fit_table(n.tab)
        del n
        gc.collect()
