"""One process plays the W ranks of a one-process-per-GPU job in turn (a measurement aid - no part of the product): every pbn_score_batch
of a device score is made W times through the library's own sharding (pbn_scoredata_set_comm, csrc/shard.hip), rank r = 0 .. W-1, with an
all-gather callback that records when the rank reached the collective (= the time of its share) and what it contributed; ranks
0 .. W-2 are stopped there (the callback fails the collective, the call returns an error that is dropped), rank W-1 receives everybody's
contribution and finishes the batch.  What a rank evaluated stays in the engine's caches as it would on its own GPU's process - a later rank
never needs it, the plan deals every unit to exactly one rank.

    with EmulatedRanks(8) as em: ...run a search...; em.batches -> [[seconds of rank 0's share, ...], ...] per batch
An N-rank job waits per batch for its slowest share and repeats the unsharded work: T_N ~ (wall - sum of all shares) + sum over batches of
the slowest share.  An ESTIMATE: no collective latency, one process's caches."""
import ctypes as C
import time

import numpy as np

from pybnesian_amd import _lib, scores


class EmulatedRanks:
    def __init__(self, world):
        self.world, self.batches, self._orig = int(world), [], None

    def __enter__(self):
        em = self
        self._orig = scores._DeviceScore._batch_raw

        def batch_raw(score, model, var, ntype, off, par, kind):
            n = len(var)
            out = np.zeros(n)
            if n == 0:
                return out
            lib = _lib.load()
            params = score._batch_params(model)
            args = (score._handle, kind, n, _lib.int_array(var), _lib.int_array(ntype), _lib.int_array(off), _lib.int_array(par if par else [0]),
                    _lib.dptr(params) if params.size else None, int(params.size), _lib.dptr(out))
            sent, times = [], []
            try:
                for r in range(em.world):
                    last = r == em.world - 1
                    t0 = time.perf_counter()

                    def gather(_u, send, count, recv, last=last, t0=t0):
                        times.append(time.perf_counter() - t0)
                        sent.append(np.ctypeslib.as_array(send, shape=(int(count),)).copy())
                        if not last:
                            return 1                     # this rank stops at the collective
                        allv = np.concatenate(sent)
                        C.memmove(recv, allv.ctypes.data, allv.nbytes)
                        return 0

                    cb = _lib.ALLGATHER_FN(gather)
                    cm = _lib.Comm(r, em.world, cb, None)
                    _lib.check(lib.pbn_scoredata_set_comm(score._handle, C.byref(cm)))
                    rc = lib.pbn_score_batch(*args)
                    if last or len(times) <= r:          # the last rank's result counts; so does a failure BEFORE the collective on any rank
                        _lib.check(rc)
            finally:
                _lib.check(lib.pbn_scoredata_set_comm(score._handle, None))
            em.batches.append(times)
            return out

        scores._DeviceScore._batch_raw = batch_raw
        return self

    def __exit__(self, *exc):
        scores._DeviceScore._batch_raw = self._orig
        return False

    def estimate(self, one_process_s):
        shares = sum(sum(t) for t in self.batches)
        slow = sum(max(t) for t in self.batches)
        return {"shares_s": shares, "slowest_sum_s": slow, "per_rank_s": one_process_s - shares + slow,
                "slowest_over_mean_share": slow / (shares / self.world) if shares else None}
