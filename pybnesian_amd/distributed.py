"""Sharding of a score batch over the GPUs of one node (one process per GPU, torch.distributed with the
"nccl" backend = RCCL over xGMI; "gloo" on CPU for tests).

Independent units = candidates: every rank holds the whole table, scores its share of the candidates
(dealt by variable set, see sharded_batch) on its own GPU, and one all_gather of <= n^2 doubles per batch gives every
rank the full result in fixed rank order, so that all ranks take the same deterministic find_max decision.
No other collective is on the data path.
"""
import numpy as np


_EMULATED = None   # measurement aid (tools/scale_emulate.py): an object with get_rank / get_world_size / emulate(times per rank)


def _dist():
    if _EMULATED is not None:
        return _EMULATED
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


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
    if have is None or have[0].numel() < cap:
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


# relative cost of one CKDE candidate by the number of variables of its joint set (variable + parents): measured sweep times at
# 1e6 x 1e5 rows (profiles/r2/sweep_dims.txt: KDE of d dimensions = the joint term; KDE of d - 1 = the marginal term)
_KDE_MS = {1: 15.5, 2: 11.1, 3: 12.8, 4: 19.7, 5: 36.3, 6: 45.4, 7: 50.7}


# batches with fewer unknown terms than this many per rank are dealt by (term, fold) instead of by term (sharded_batch)
_SPLIT_TERMS_BELOW = 4


def _kde_cost(d):
    return _KDE_MS.get(d, 51.6 + 5.0 * max(0, d - 8)) if d >= 1 else 0.0


def deal_sets(keys, world):
    """Owner rank of every variable set: longest processing time first on a cost model (one joint sweep per set + one marginal
    sweep per candidate, x folds x rows being common factors), ties by a hash of the set, each set to the least loaded rank
    (lowest rank on ties).  Deterministic, identical on every rank.  keys: list of (set, number of candidates) in order of
    first appearance."""
    # equal costs (the initial cache: every set a pair) are ordered by a hash of the set, not by appearance: in order of appearance
    # rank r gets the sets i = r mod world, i.e. the pairs of the SAME few variables - and a variable whose sweeps prune badly made
    # its rank 25 % slower than the mean of eight (tools/scale_emulate.py); scattered, the ranks' sums differ by a few percent
    def mix(key):
        h = 0x9E3779B9
        for v in key:
            h = ((h ^ (int(v) + 0x7F4A7C15)) * 0x85EBCA6B) & 0xFFFFFFFF
            h ^= h >> 13
        return h

    cost = [(_kde_cost(len(k)) + n * _kde_cost(len(k) - 1), mix(k), -i) for i, (k, n) in enumerate(keys)]
    order = sorted(range(len(keys)), key=lambda i: cost[i], reverse=True)
    load = [0.0] * world
    owner = [0] * len(keys)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += cost[i][0]
    return owner


def _raise_if_failed(flags, failure, where):
    """Every rank took part in the collective; if any of them failed while computing its share, all raise together."""
    bad = [r for r, f in enumerate(flags) if f != 0.0]
    if failure is not None:
        raise failure
    if bad:
        raise RuntimeError(f"{where}: rank(s) {bad} failed while computing their share of the batch")


def shard_indices(n, rank, world):
    return list(range(rank, n, world))


def _gather_shares(dist, world, rank, lists, compute, where):
    """Rank r evaluates compute(lists[r]) -> values; every rank gets every rank's values (one all_gather; a failure anywhere raises
    everywhere).  Under the emulation hook one process plays the ranks in turn and reports each share's seconds."""
    if hasattr(dist, "emulate"):
        import time

        times, vals = [], []
        for r in range(world):
            t0 = time.perf_counter()
            vals.append(np.asarray(compute(lists[r]), dtype=np.float64) if lists[r] else np.zeros(0))
            times.append(time.perf_counter() - t0)
        dist.emulate(times, [len(l) for l in lists])
        return vals
    mine = lists[rank]
    per = max(1, max(len(l) for l in lists))
    buf = np.zeros(per + 1)                    # last slot: this rank's error flag
    failure = None
    try:
        buf[: len(mine)] = compute(mine) if mine else np.zeros(0)
    except Exception as ex:                    # never skip the collective: the other ranks are already on their way to it
        failure = ex
        buf[:] = np.nan
        buf[per] = 1.0
    allv = _all_gather(dist, buf).reshape(world, per + 1)
    _raise_if_failed(allv[:, per], failure, where)
    return [allv[r, : len(lists[r])] for r in range(world)]


def _deal(keys, world):
    owner = deal_sets(keys, world)
    return [[i for i in range(len(keys)) if owner[i] == r] for r in range(world)]


def sharded_batch(score, model, var, ntype, off, par, kind, shard_all=False):
    """Score a batch; with torch.distributed initialised the device-heavy work (CKDE node type under a likelihood score: k sweeps
    per term) is sharded over the ranks, while LinearGaussian candidates - O(p^3) host arithmetic on replicated moments - are computed
    redundantly by every rank (sharding them buys nothing and costs a collective).
    Continuous CKDE candidates are sharded by TERM: local(v | P) = A({v} u P) - A(P), and A({s}) serves every child of s, A({s, t})
    both directions of the arc - dealing candidates made every rank sweep all 64 single-variable terms of the initial cache itself
    (150 of its 535 ms at eight ranks, tools/scale_emulate.py).  The unknown terms of the batch are dealt by cost, evaluated
    (pbn_score_terms), all-gathered and installed on every rank (pbn_score_terms_put); every rank then assembles the candidates from
    the same doubles - the very sums the one-process run forms.  Candidates with discrete parents (hybrid scores) are dealt whole."""
    from . import _lib

    dist = _dist()
    n = len(var)
    if dist is None or n == 0:
        return score._batch_raw(model, var, ntype, off, par, kind)
    heavy = [i for i in range(n) if shard_all or ntype[i] == _lib.PBN_NODE_CKDE]
    out = np.zeros(n)

    def sub(idx):
        o, p = [0], []
        for i in idx:
            p.extend(par[off[i]: off[i + 1]])
            o.append(len(p))
        return score._batch_raw(model, [var[i] for i in idx], [ntype[i] for i in idx], o, p, kind)

    rank, world = dist.get_rank(), dist.get_world_size()
    n_cont = len(score._table.names) if getattr(score, "_table", None) is not None else None   # column ids below it are continuous
    by_term = [i for i in heavy if ntype[i] == _lib.PBN_NODE_CKDE and kind in (_lib.PBN_SCORE_CVLIK, _lib.PBN_SCORE_HOLDOUT) and n_cont is not None
               and var[i] < n_cont and all(q < n_cont for q in par[off[i]: off[i + 1]]) and hasattr(score, "_terms")]
    if len(by_term) >= 2:
        terms, seen = [], {}
        for i in by_term:
            ps = list(par[off[i]: off[i + 1]])
            d = len(ps) + 1
            for key in ((d,) + tuple(sorted([var[i]] + ps)), ((d,) + tuple(sorted(ps))) if ps else None):
                if key is not None and key not in seen:
                    seen[key] = len(terms)
                    terms.append(key)
        missing = score._terms("missing", kind, terms)
        todo = [t for t, mflag in zip(terms, missing) if mflag]
        regions = score._term_regions(kind) if hasattr(score, "_term_regions") else 1
        if todo and regions > 1 and len(todo) < _SPLIT_TERMS_BELOW * world:
            # an update batch of a search: a handful of terms, fewer than ranks or not many more - dealt whole, five terms leave three
            # of eight ranks idle and two with double work.  Dealt (term, fold) by (term, fold) instead: a term's total is its folds
            # added in fold order, so the per-fold values are gathered and every rank adds them in that order - the one-process double.
            items = [(j, f) for j in range(len(todo)) for f in range(regions)]
            cost = [_kde_cost(len(todo[j]) - 1) for j, _ in items]
            order = sorted(range(len(items)), key=lambda i: (-cost[i], i))
            load, lists = [0.0] * world, [[] for _ in range(world)]
            for i in order:
                r = min(range(world), key=lambda q: (load[q], q))
                lists[r].append(i)
                load[r] += cost[i]
            vals = _gather_shares(dist, world, rank, lists, lambda idx: score._terms("eval_regions", kind, [todo[items[i][0]] for i in idx],
                                                                                     regions=[items[i][1] for i in idx]), "sharded_batch")
            per = np.zeros((len(todo), regions))
            for r in range(world):
                for i, v in zip(lists[r], np.asarray(vals[r], dtype=np.float64)):
                    per[items[i][0], items[i][1]] = v
            totals = []
            for j in range(len(todo)):
                acc = 0.0
                for x in per[j].tolist():   # the folds in order, as the engine adds them
                    acc += x
                totals.append(acc)
            score._terms("put", kind, todo, np.asarray(totals))
        elif todo:
            lists = _deal([(t[1:], 0) for t in todo], world)
            vals = _gather_shares(dist, world, rank, lists, lambda idx: score._terms("eval", kind, [todo[j] for j in idx]), "sharded_batch")
            flat_t = [todo[j] for r in range(world) for j in lists[r]]
            flat_v = np.concatenate([np.asarray(v, dtype=np.float64) for v in vals]) if flat_t else np.zeros(0)
            score._terms("put", kind, flat_t, flat_v)
        taken = set(by_term)
        heavy = [i for i in heavy if i not in taken]
    else:
        by_term = []
    whole = set(heavy)
    rest = [i for i in range(n) if i not in whole]   # light candidates + the term-sharded ones: every rank, from the shared terms
    if rest:
        out[rest] = sub(rest)
    # CKDE candidates with discrete parents: the update batches of a restricted search hold a handful of them - fewer than ranks, each
    # 10-50 ms of sweeps - so the ranks share every candidate's SLICES (configuration x fold): rank r evaluates the parts dealt to it
    # (by cost, identically on every rank) of the engine's 64 fixed parts (pbn_score_batch_parts), the per-part sums are all-gathered, added over the ranks (a part
    # is non-zero on one rank only) and then over the parts in order - the one-process sum, bit for bit (tools/scale_emulate.py on
    # BASELINE config 5, eight ranks: 3.1 s dealing whole candidates, a third of the batches unsharded for holding one candidate)
    sliced = [i for i in heavy if hasattr(score, "_batch_parts") and world <= 64 and ntype[i] == _lib.PBN_NODE_CKDE and n_cont is not None
              and kind in (_lib.PBN_SCORE_CVLIK, _lib.PBN_SCORE_HOLDOUT) and var[i] < n_cont and any(q >= n_cont for q in par[off[i]: off[i + 1]])]
    if sliced:
        o, p = [0], []
        for i in sliced:
            p.extend(par[off[i]: off[i + 1]])
            o.append(len(p))
        sv, st = [var[i] for i in sliced], [ntype[i] for i in sliced]
        share = lambda r: score._batch_parts(model, sv, st, o, p, kind, r, world).reshape(-1)
        if hasattr(dist, "emulate"):
            import time

            times, total = [], np.zeros(len(sliced) * 64)
            for r in range(world):
                t0 = time.perf_counter()
                total += share(r)
                times.append(time.perf_counter() - t0)
            dist.emulate(times, [len(sliced)] * world)
        else:
            buf = np.zeros(len(sliced) * 64 + 1)
            failure = None
            try:
                buf[:-1] = share(rank)
            except Exception as ex:                # never skip the collective
                failure = ex
                buf[:] = np.nan
                buf[-1] = 1.0
            allv = _all_gather(dist, buf).reshape(world, -1)
            _raise_if_failed(allv[:, -1], failure, "sharded_batch")
            total = np.zeros(len(sliced) * 64)
            for r in range(world):                 # exact: every part is non-zero on one rank only
                total += allv[r, :-1]
        total = total.reshape(len(sliced), 64)
        for j, i in enumerate(sliced):
            acc = 0.0
            for x in total[j].tolist():            # the parts in order, as the engine adds them
                acc += x
            out[i] = acc
        done = set(sliced)
        heavy = [i for i in heavy if i not in done]
    if len(heavy) < 2:
        if heavy:
            out[heavy] = sub(heavy)
        return out
    # whole candidates (anything else that is heavy): those over the same variable set share their sums in the engine's set-function
    # cache and go to the same rank; the sets are dealt by cost (deal_sets), identically on every rank
    set_of, counts = {}, []
    for i in heavy:
        key = tuple(sorted([var[i]] + list(par[off[i]: off[i + 1]])))
        if key not in set_of:
            set_of[key] = len(set_of)
            counts.append([key, 0])
        counts[set_of[key]][1] += 1
    set_owner = deal_sets([(k, c) for k, c in counts], world)
    owner = [set_owner[set_of[tuple(sorted([var[i]] + list(par[off[i]: off[i + 1]])))]] for i in heavy]
    lists = [[heavy[j] for j in range(len(heavy)) if owner[j] == r] for r in range(world)]
    vals = _gather_shares(dist, world, rank, lists, sub, "sharded_batch")
    for r in range(world):
        out[lists[r]] = vals[r]
    return out


def sharded_slogl(factor, df):
    """KDE / ProductKDE / CKDE `slogl(df)` with the test rows split over the ranks (independent units =
    test rows, fitted model replicated).  Every rank passes the same `df`; rank r evaluates the contiguous slice
    [r*m/W, (r+1)*m/W), the W partial sums are all-gathered and added in rank order (deterministic), and every rank
    returns the same total.  Without torch.distributed it is `factor.slogl(df)`."""
    dist = _dist()
    if dist is None:
        return factor.slogl(df)
    import torch

    from .dataset import as_record_batch

    rb = as_record_batch(df)
    rank, world = dist.get_rank(), dist.get_world_size()
    m = rb.num_rows
    lo, hi = (m * rank) // world, (m * (rank + 1)) // world
    buf = np.zeros(2)
    failure = None
    try:
        buf[0] = factor.slogl(rb.slice(lo, hi - lo)) if hi > lo else 0.0
    except Exception as ex:
        failure = ex
        buf[:] = (np.nan, 1.0)
    allv = _all_gather(dist, buf).reshape(world, 2)
    _raise_if_failed(allv[:, 1], failure, "sharded_slogl")
    total = 0.0
    for v in allv[:, 0].tolist():
        total += v
    return total


def reduce_moments(handle):
    """Row-sharded Gram (BGe / BIC / LG-CV row): every rank computed the moments of its share of
    each region (pbn_scoredata_create_sharded); all-gather the (k+1) * (n + n^2) doubles, add them in rank order on
    the host (deterministic, identical on every rank) and install the totals."""
    import ctypes as C

    import torch

    from . import _lib

    dist = _dist()
    lib = _lib.load()
    ln = C.c_int64(0)
    _lib.check(lib.pbn_scoredata_moments(handle, None, C.byref(ln), 0))
    buf = np.zeros(ln.value)
    _lib.check(lib.pbn_scoredata_moments(handle, _lib.dptr(buf), C.byref(ln), 0))
    if dist is not None:
        world = dist.get_world_size()
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        send = torch.from_numpy(buf).to(dev)
        recv = torch.empty(world * buf.size, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send)
        parts = recv.cpu().numpy().reshape(world, buf.size)
        buf = parts[0].copy()
        for r in range(1, world):
            buf += parts[r]
    _lib.check(lib.pbn_scoredata_moments(handle, _lib.dptr(buf), C.byref(ln), 1))


def sharded_ci_batch(fn, native_batch, user, errors):
    """Independence tests of one MMPC step spread over the ranks (same idea as the delta cache: independent units, one
    all_gather per batch).  Every rank runs the same search; test i of a batch is evaluated by rank i % world - through
    the native batched callback when the test has one, else one by one - and the p-values are gathered, so every rank
    sees bit-identical numbers and takes identical decisions.  Returns (batch callback or None, keep-alive)."""
    import ctypes as C

    from . import _lib

    dist = _dist()
    if dist is None:
        return native_batch, None
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    single = fn if callable(fn) else _lib.CI_PVALUE_FN(fn.value)
    native = _lib.CI_BATCH_FN(native_batch.value) if native_batch is not None else None
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

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
