"""One process per GPU (torch.distributed: "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU for tests): the thin caller of the
library's sharded delta cache.

The planning - which rank evaluates which CKDE term, (term, fold) pair, hybrid slice part or whole candidate - the evaluation of a
rank's share and the assembly of every candidate from the gathered doubles live BEHIND the C ABI (csrc/shard.hip: pbn_scoredata_set_comm,
pbn_shard_batch, pbn_scoredata_reduce_moments, pbn_kde_slogl_sharded; include/pbn_hip.h "one process per GPU").  This module only supplies
the ONE collective the library asks its host for: an all-gather of doubles, here through torch.distributed.  A C++ host supplies
ncclAllGather instead and needs nothing of this file.

Independent units = candidates / terms / slices / test rows: every rank holds the whole table, one all_gather of a few
hundred doubles per batch gives every rank the full result in fixed rank order, so that all ranks take the same deterministic find_max
decision.  No other collective is on the data path.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

# The collective path normally needs world_size > 1.  FORCE (or PBN_FORCE_DIST=1) takes it at world size 1 as well - one rank dealing
# everything to itself through a real all_gather: how a one-GPU box executes the RCCL path (tests/test_distributed_gpu.py, bench.py).
FORCE = False


def _forced():
    return FORCE or os.environ.get("PBN_FORCE_DIST", "") not in ("", "0")


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist if (dist.get_world_size() > 1 or _forced()) else None


_GATHER_BUFS = {}


def _all_gather(dist, buf):
    """all_gather of one float64 vector per rank -> (world * len) numpy array.  nccl (RCCL): through persistent pinned staging
    and device buffers with ONE stream synchronisation (no allocation, no pageable copy per batch); gloo: host tensors."""
    import torch

    buf = np.ascontiguousarray(buf, dtype=np.float64)
    world = dist.get_world_size()
    if dist.get_backend() != "nccl":
        send = torch.from_numpy(buf)
        recv = torch.empty(world * send.numel(), dtype=torch.float64)
        dist.all_gather_into_tensor(recv, send)
        return recv.numpy()
    dev = torch.device("cuda", torch.cuda.current_device())
    # ONE set of staging buffers per device, grown to the next power of two and sliced for the current length (batch lengths vary over
    # a search: a set per exact length would accumulate pinned allocations, each costing more than the pageable copy it replaces)
    cap = 1 << max(10, int(buf.size - 1).bit_length())
    have = _GATHER_BUFS.get(dev.index)
    if have is None or have[0].numel() < cap or have[2].numel() < world * cap:
        have = (torch.empty(cap, dtype=torch.float64).pin_memory(), torch.empty(cap, dtype=torch.float64, device=dev),
                torch.empty(world * cap, dtype=torch.float64, device=dev), torch.empty(world * cap, dtype=torch.float64).pin_memory())
        _GATHER_BUFS[dev.index] = have
    n = buf.size
    send_h, send_d, recv_d, recv_h = have[0][:n], have[1][:n], have[2][: world * n], have[3][: world * n]
    send_h.numpy()[:] = buf
    send_d.copy_(send_h, non_blocking=True)
    dist.all_gather_into_tensor(recv_d, send_d)
    recv_h.copy_(recv_d, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return recv_h.numpy().copy()


class Comm:
    """pbn_comm over a torch.distributed process group: rank, world and the all-gather callback (kept alive with the object).  An
    exception inside the collective is kept in `errors` and re-raised by `check` once the C call has returned."""

    def __init__(self, dist):
        self.dist, self.errors, self.gathers = dist, [], 0

        def gather(_user, send, count, recv):
            try:
                allv = _all_gather(dist, np.ctypeslib.as_array(send, shape=(int(count),)))
                C.memmove(recv, allv.ctypes.data, allv.nbytes)
                self.gathers += 1
                return 0
            except Exception as ex:  # surfaced after the C call returns
                self.errors.append(ex)
                return 1

        self._cb = _lib.ALLGATHER_FN(gather)
        self.struct = _lib.Comm(int(dist.get_rank()), int(dist.get_world_size()), self._cb, None)

    def ref(self):
        return C.byref(self.struct)

    def check(self, rc):
        if self.errors:
            err, self.errors = self.errors[0], []
            raise err
        _lib.check(rc)


_COMM = None


def comm():
    """The process's communicator, or None without torch.distributed (or at world size 1 unless forced)."""
    global _COMM
    dist = _dist()
    if dist is None:
        return None
    if _COMM is None or _COMM.dist is not dist or _COMM.struct.world != dist.get_world_size() or _COMM.struct.rank != dist.get_rank():
        _COMM = Comm(dist)
    return _COMM


def collective_calls():
    """All-gather calls the process's communicator has made so far (0 without one): what the tests count."""
    return 0 if _COMM is None else _COMM.gathers


def deal_sets(keys, world, rows=(1_000_000, 100_000), regions=1):
    """Owner rank of every variable set (pbn_shard_deal on pbn_shard_term_cost: one joint sweep per set + one marginal sweep per
    candidate; longest processing time first, equal costs by a hash of the set, each set to the least loaded rank).  Deterministic,
    identical on every rank.  keys: list of (set, number of candidates)."""
    lib = _lib.load()

    def mix(key):
        h = 0x9E3779B9
        for v in key:
            h = ((h ^ (int(v) + 0x7F4A7C15)) * 0x85EBCA6B) & 0xFFFFFFFF
            h ^= h >> 13
        return h

    n = len(keys)
    cost = np.asarray([regions * (term_cost(len(k), *rows) + c * term_cost(len(k) - 1, *rows)) for k, c in keys], dtype=np.float64)
    tie = (C.c_uint32 * max(1, n))(*[mix(k) for k, _ in keys])
    owner = (C.c_int * max(1, n))()
    _lib.check(lib.pbn_shard_deal(n, _lib.dptr(cost) if n else None, tie, int(world), None, owner))
    return list(owner)[:n]


def term_cost(dims, train_rows=1_000_000, test_rows=100_000):
    return float(_lib.load().pbn_shard_term_cost(int(dims), int(train_rows), int(test_rows)))


def shard_indices(n, rank, world):
    return list(range(rank, n, world))


class _PyEngine:
    """pbn_shard_engine over a Python score object (`_batch_raw`, optionally `_terms` / `_term_regions` / `_batch_parts` / `_shard_shape`):
    scores that are not bound to the library's own handle - user-defined ones, the stand-ins of the CPU tests."""

    def __init__(self, score, model):
        self.errors = []
        ip, dp = _lib._ip, _lib._dp
        n_cont = len(score._table.names) if getattr(score, "_table", None) is not None else None

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as ex:  # surfaced after the C call returns
                    self.errors.append(ex)
                    return 1
            return wrapped

        def cands(n, var, nt, off, par):
            o = [off[i] for i in range(n + 1)]
            return [var[i] for i in range(n)], [nt[i] for i in range(n)], o, [par[i] for i in range(o[-1])]

        def terms_of(n, off, vars_, m):
            return [(m[i],) + tuple(vars_[j] for j in range(off[i], off[i + 1])) for i in range(n)]

        def shape(_u, kind, regions, ntr, nte):
            regions[0] = int(score._term_regions(kind)) if hasattr(score, "_term_regions") else 1
            rows = score._shard_shape(kind) if hasattr(score, "_shard_shape") else (0, 0)
            ntr[0], nte[0] = int(rows[0]), int(rows[1])

        def batch(_u, kind, n, var, nt, off, par, out):
            v, t, o, p = cands(n, var, nt, off, par)
            res = score._batch_raw(model, v, t, o, p, kind)
            for i in range(n):
                out[i] = res[i]

        def missing(_u, kind, n, off, vars_, m, flags):
            res = score._terms("missing", kind, terms_of(n, off, vars_, m))
            for i in range(n):
                flags[i] = int(res[i])

        def terms(_u, kind, n, off, vars_, m, out):
            res = score._terms("eval", kind, terms_of(n, off, vars_, m))
            for i in range(n):
                out[i] = res[i]

        def term_regions(_u, kind, n, off, vars_, m, region, out):
            res = score._terms("eval_regions", kind, terms_of(n, off, vars_, m), regions=[region[i] for i in range(n)])
            for i in range(n):
                out[i] = res[i]

        def put(_u, kind, n, off, vars_, m, values):
            score._terms("put", kind, terms_of(n, off, vars_, m), np.asarray([values[i] for i in range(n)], dtype=np.float64))

        def parts(_u, kind, n, var, nt, off, par, part, n_parts, out):
            v, t, o, p = cands(n, var, nt, off, par)
            res = np.asarray(score._batch_parts(model, v, t, o, p, kind, part, n_parts), dtype=np.float64).reshape(-1)
            for i in range(n * 64):
                out[i] = res[i]

        E = _lib.ShardEngine
        f = dict(E._fields_)
        has_terms = n_cont is not None and hasattr(score, "_terms")
        has_parts = n_cont is not None and hasattr(score, "_batch_parts")
        self._cbs = dict(shape=f["shape"](guard(shape)), batch=f["batch"](guard(batch)),
                         terms_missing=f["terms_missing"](guard(missing)) if has_terms else f["terms_missing"](),
                         terms=f["terms"](guard(terms)) if has_terms else f["terms"](),
                         term_regions=f["term_regions"](guard(term_regions)) if has_terms and hasattr(score, "_term_regions") else f["term_regions"](),
                         terms_put=f["terms_put"](guard(put)) if has_terms else f["terms_put"](),
                         batch_parts=f["batch_parts"](guard(parts)) if has_parts else f["batch_parts"]())
        self.struct = E(None, int(n_cont) if n_cont is not None else 0, *[self._cbs[k] for k in ("shape", "batch", "terms_missing", "terms", "term_regions",
                                                                                                   "terms_put", "batch_parts")], f["term_price"]())


def sharded_batch(score, model, var, ntype, off, par, kind, shard_all=False):
    """Score a batch; with torch.distributed initialised the device-heavy work is sharded over the ranks by the library (shard.hip).  A device
    score created under the process group has the communicator bound to its handle: its plain `_batch_raw` (pbn_score_batch) IS the sharded
    call.  Any other score object goes through pbn_shard_batch with its Python methods as the engine."""
    cm = comm()
    n = len(var)
    bound = getattr(score, "_comm", None)
    if bound is not None and bound is not cm and getattr(score, "_handle", None) is not None:
        # the process group the score was created under is gone (destroy_process_group, or another group / world size since): the handle
        # still holds that communicator's rank, world and callback.  Its moments were reduced over the old group and are complete, so the
        # handle is re-bound to the current communicator - or un-bound (one process) - instead of running a dead closure
        if bound.errors:
            err, bound.errors = bound.errors[0], []
            raise err
        _lib.check(_lib.load().pbn_scoredata_set_comm(score._handle, cm.ref() if cm is not None else None))
        score._comm = cm
    if cm is None or n == 0:
        return score._batch_raw(model, var, ntype, off, par, kind)
    if getattr(score, "_comm", None) is cm and not shard_all:
        try:
            return score._batch_raw(model, var, ntype, off, par, kind)
        finally:
            if cm.errors:   # the collective itself failed inside the library call: its own exception
                err, cm.errors = cm.errors[0], []
                raise err
    eng = _PyEngine(score, model)
    out = np.zeros(n)
    rc = _lib.load().pbn_shard_batch(C.byref(eng.struct), cm.ref(), int(kind), n, _lib.int_array(var), _lib.int_array(ntype), _lib.int_array(off),
                                     _lib.int_array(par if par else [0]), int(bool(shard_all)), _lib.dptr(out))
    if eng.errors:
        raise eng.errors[0]
    cm.check(rc)
    return out


def sharded_slogl(factor, df):
    """KDE / ProductKDE / CKDE `slogl(df)` with the test rows split over the ranks (independent units = test rows, fitted
    model replicated): pbn_kde_slogl_sharded - rank r evaluates the contiguous slice [r*m/W, (r+1)*m/W), the W partial sums are
    all-gathered and added in rank order, every rank returns the same total.  Without torch.distributed it is `factor.slogl(df)`."""
    cm = comm()
    if cm is None:
        return factor.slogl(df)
    if getattr(factor, "_split", False) or not hasattr(factor, "_upload_test") or not hasattr(factor, "_handle"):
        # factors that are not one library handle: this rank's rows through the factor's own slogl, the same gather
        from .dataset import as_record_batch

        rb = as_record_batch(df)
        rank, world, m = cm.struct.rank, cm.struct.world, rb.num_rows
        lo, hi = (m * rank) // world, (m * (rank + 1)) // world
        buf, failure = np.zeros(2), None
        try:
            buf[0] = factor.slogl(rb.slice(lo, hi - lo)) if hi > lo else 0.0
        except Exception as ex:
            failure, buf[:] = ex, (np.nan, 1.0)
        allv = _all_gather(cm.dist, buf).reshape(world, 2)
        if failure is not None:
            raise failure
        bad = [r for r in range(world) if allv[r, 1] != 0.0]
        if bad:
            raise RuntimeError(f"sharded_slogl: rank(s) {bad} failed while computing their share of the batch")
        total = 0.0
        for v in allv[:, 0].tolist():
            total += v
        return total
    _, table, _ = factor._upload_test(df)
    res = C.c_double(0.0)
    rc = _lib.load().pbn_kde_slogl_sharded(factor._handle, table.handle, _lib.int_array(table.index(factor._variables)), 0, table.num_rows,
                                           cm.ref(), C.byref(res))
    cm.check(rc)
    return res.value


def reduce_moments(handle):
    """Row-sharded Gram (BGe / BIC / LG-CV row): every rank computed the moments of its share of each region
    (pbn_scoredata_create_sharded); pbn_scoredata_reduce_moments all-gathers them, adds them in rank order (deterministic, identical on
    every rank) and installs the totals."""
    cm = comm()
    if cm is None:
        ln = C.c_int64(0)
        lib = _lib.load()
        _lib.check(lib.pbn_scoredata_moments(handle, None, C.byref(ln), 0))
        buf = np.zeros(ln.value)
        _lib.check(lib.pbn_scoredata_moments(handle, _lib.dptr(buf), C.byref(ln), 0))
        _lib.check(lib.pbn_scoredata_moments(handle, _lib.dptr(buf), C.byref(ln), 1))
        return
    cm.check(_lib.load().pbn_scoredata_reduce_moments(handle, cm.ref()))


def sharded_ci_batch(fn, native_batch, user, errors):
    """Independence tests of one MMPC step spread over the ranks (same idea as the delta cache: independent units, one
    all_gather per batch).  Every rank runs the same search; test i of a batch is evaluated by rank i % world - through
    the native batched callback when the test has one, else one by one - and the p-values are gathered, so every rank
    sees bit-identical numbers and takes identical decisions.  Returns (batch callback or None, keep-alive)."""
    cm = comm()
    if cm is None:
        return native_batch, None
    dist = cm.dist
    rank, world = cm.struct.rank, cm.struct.world
    single = fn if callable(fn) else _lib.CI_PVALUE_FN(fn.value)
    native = _lib.CI_BATCH_FN(native_batch.value) if native_batch is not None else None

    def batch(_user, n, v1, v2, off, cond, out):
        mine = list(range(rank, n, world))
        per = (n + world - 1) // world
        buf = np.full(per + 1, np.nan)
        buf[per] = 0.0                               # error flag
        try:
            local = np.full(len(mine), np.nan)
            if mine:
                a = _lib.int_array([v1[i] for i in mine])
                b = _lib.int_array([v2[i] for i in mine])
                o, c = [0], []
                for i in mine:
                    c.extend(cond[j] for j in range(off[i], off[i + 1]))
                    o.append(len(c))
                if native is not None:
                    native(user, len(mine), a, b, _lib.int_array(o), _lib.int_array(c or [0]), _lib.dptr(local))
                else:
                    for q, i in enumerate(mine):
                        ci = _lib.int_array(c[o[q]: o[q + 1]] or [0])
                        local[q] = single(user, v1[i], v2[i], o[q + 1] - o[q], ci)
            buf[: len(mine)] = local
        except Exception as ex:  # surfaced after the C call returns; the collective below still happens
            errors.append(ex)
            buf[per] = 1.0
        try:
            allv = _all_gather(dist, buf).reshape(world, per + 1)
            if np.any(allv[:, per] != 0.0) and not errors:
                errors.append(RuntimeError("sharded_ci_batch: another rank failed while computing its share of the tests"))
            for r in range(world):
                idx = range(r, n, world)
                for q, i in enumerate(idx):
                    out[i] = allv[r, q] if not errors else float("nan")
        except Exception as ex:
            errors.append(ex)
            for i in range(n):
                out[i] = float("nan")

    cb = _lib.CI_BATCH_FN(batch)
    return C.cast(cb, C.c_void_p), (cb, single, native)
