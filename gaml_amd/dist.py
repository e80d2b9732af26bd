"""One process per GPU: the collectives around a sharded likelihood evaluation.

The C ABI knows nothing about process groups; this module is the plumbing a multi-GPU caller needs,
written once over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm):

  cold path only   all-reduce(max) of the largest record position of every newly aligned window
                   (the reference's `max_pos - 5` filter, graph.cc:577, is a maximum over ALL reads)
  penalty > 0 only all-gather of the per-rank coverage maps of each paired set, then every rank
                   sweeps the union (bad_bases, graph.cc:1893-1919)
  every step       ONE all-reduce(sum) of 4 f64 per read set {sum of logs, floored reads, bad_bases, reads}
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import api


class ShardedScorer:
    """CalcProb over a context that holds one shard of every read set. All ranks call calc_prob with
    the same paths and get the same value."""

    def __init__(self, ctx: api.Context, group=None, stream: "torch.cuda.Stream | None" = None, host_exchange: "str | None" = None):
        """host_exchange: name of a POSIX shared-memory block ("/gaml_<something>", the same on every rank) when all
        ranks run on ONE node and no read set has a coverage penalty: each step is then a blocking evaluation (partials
        land in host memory) + a ~1 us sum through shared memory, instead of finisher kernel -> RCCL all-reduce of 32
        bytes -> fetch (about 30 us of dependent dispatches). RCCL stays the exchange for everything else: window
        maxima on the cold path, coverage maps, PacBio events -- and for the partials when host_exchange is None."""
        self.ctx = ctx
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # a real (non-null) HIP stream: kernels, collectives and the D2H copy are ordered on it
        self.stream = stream or torch.cuda.Stream()
        self.d_part = torch.zeros(4 * max(1, ctx.num_readsets()), dtype=torch.float64, device="cuda")
        self.h_part = torch.zeros(4 * max(1, ctx.num_readsets()), dtype=torch.float64).pin_memory()
        self._h_np = self.h_part.numpy()
        self._h_ptr = self.h_part.data_ptr()
        self._d_ptr = self.d_part.data_ptr()
        self._has_pacbio = any(ctx.readset_kind(i) == 2 for i in range(ctx.num_readsets()))
        self._maps = None
        self._gathered = None
        # RCCL works on device tensors; gloo (tests: several ranks sharing one GPU, or CPU-only collectives)
        # does not support every collective on them, so there the buffers take a detour through the host
        self._host_collectives = dist.get_backend(group) == "gloo"
        self._n_part = self.d_part.numel()
        self._host_exchange = host_exchange
        if host_exchange:
            if self.rank == 0:
                ctx.shm_exchange_open(host_exchange, 0, self.world, self._n_part)  # creates a fresh block
            dist.barrier(group=group)
            if self.rank != 0:
                ctx.shm_exchange_open(host_exchange, self.rank, self.world, self._n_part)
            dist.barrier(group=group)  # everybody has mapped the block before anybody publishes

    def close(self):
        if self._host_exchange:
            dist.barrier(group=self.group)
            self.ctx.shm_exchange_close(unlink_name=self.rank == 0)
            self._host_exchange = None

    def _score_host(self, fp):
        """Blocking evaluation + shared-memory sum (single node, no coverage penalty)."""
        ctx = self.ctx
        pending, total_len = ctx.eval_begin_fast(fp) if isinstance(fp, api.FlatPaths) else ctx.eval_begin(fp)
        if pending:  # cold path: new windows were aligned -- their largest positions are a maximum over all ranks' reads
            mx = torch.from_numpy(ctx.eval_pending_maxpos().copy()).cuda()
            self._all_reduce(mx, dist.ReduceOp.MAX)
            ctx.eval_apply_maxpos(mx.cpu().numpy())
        ctx.eval_finish_fast(self._h_ptr)
        ctx.shm_allreduce_sum(self._h_ptr, self._n_part)
        return total_len

    def _fetch(self):
        """Reduced partials -> self.h_part: a one-block kernel publishes them in mapped pinned memory behind the
        all-reduce on the stream, the host polls (gaml_hip_fetch_async / _wait) -- no D2H copy command, no
        stream-synchronize wake-up (10 us less per step than `h_part.copy_(d_part); stream.synchronize()`)."""
        self.ctx.fetch_async(self._d_ptr, self._n_part, self.stream.cuda_stream)
        self.ctx.fetch_wait(self._h_ptr, self._n_part)

    def _all_reduce(self, t, op):
        if self._host_collectives:
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    def _all_gather(self, out, own):
        if self._host_collectives:
            parts = [torch.empty(own.numel(), dtype=own.dtype) for _ in range(self.world)]
            dist.all_gather(parts, own.cpu(), group=self.group)
            out.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(out, own, group=self.group)

    def _enqueue(self, paths, d_part):
        """Everything of one evaluation up to (not including) the all-reduce of its partials."""
        ctx = self.ctx
        pending, total_len = ctx.eval_begin_fast(paths) if isinstance(paths, api.FlatPaths) else ctx.eval_begin(paths)
        if pending:
            mx = torch.from_numpy(ctx.eval_pending_maxpos().copy()).cuda()
            self._all_reduce(mx, dist.ReduceOp.MAX)
            ctx.eval_apply_maxpos(mx.cpu().numpy())
        sp = self.stream.cuda_stream
        n_maps = ctx.eval_score_async(d_part.data_ptr(), sp)
        for i in range(n_maps):
            nbytes = ctx.eval_coverage_bytes(i)
            if self._maps is None or self._maps.numel() < nbytes:
                self._maps = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                self._gathered = torch.empty(nbytes * self.world, dtype=torch.uint8, device="cuda")
            own = self._maps[:nbytes]
            gathered = self._gathered[: nbytes * self.world]
            ctx.eval_coverage_export_async(i, own.data_ptr(), nbytes, sp)
            self._all_gather(gathered, own)
            ctx.eval_coverage_finish_async(i, gathered.data_ptr(), self.world, self.rank == 0, sp)
        for i in range(ctx.eval_pacbio_pending() if self._has_pacbio else 0):  # PacBio sets with a penalty: alignment intervals of all ranks (device lists)
            n_own = ctx.eval_pacbio_intervals(i)
            sizes = torch.zeros(self.world, dtype=torch.int64, device="cuda")
            sizes[self.rank] = n_own
            self._all_reduce(sizes, dist.ReduceOp.SUM)
            sizes = sizes.cpu().tolist()
            width = max(1, max(sizes))  # intervals per rank in the gather; 4 int32 each
            own = torch.zeros(4 * width, dtype=torch.int32, device="cuda")
            ctx.eval_pacbio_export_async(i, own.data_ptr(), width, sp)
            gathered = torch.empty(self.world * 4 * width, dtype=torch.int32, device="cuda")
            self._all_gather(gathered, own)
            rows = gathered.view(self.world, 4 * width)
            merged = torch.cat([rows[r, : 4 * sizes[r]] for r in range(self.world)]).contiguous()  # rank order, the padding dropped
            ctx.eval_pacbio_finish_async(i, merged.data_ptr(), sum(sizes), self.rank == 0, sp)
            self._keep = (own, gathered, merged)  # (alive until the stream has run the sweep)
        return total_len

    def calc_prob(self, paths):
        """Blocking, like CalcProb. (Callers in a tight loop may wrap the loop in `with torch.cuda.stream(
        scorer.stream)` themselves: the context manager costs a few microseconds per entry.)"""
        if self._host_exchange:
            total_len = self._score_host(paths)
            prob = self.ctx.combine_fast(self._h_ptr, total_len)
            return prob, self.ctx.last_zeros, total_len
        if torch.cuda.current_stream() != self.stream:
            with torch.cuda.stream(self.stream):
                return self.calc_prob(paths)
        total_len = self._enqueue(paths, self.d_part)
        self._all_reduce(self.d_part, dist.ReduceOp.SUM)  # the one collective of the hot path
        self._fetch()
        prob = self.ctx.combine_fast(self._h_ptr, total_len)
        return prob, self.ctx.last_zeros, total_len

    def score(self, fp: "api.FlatPaths") -> float:
        """calc_prob for tight loops: prebuilt FlatPaths in, the value out (floored counts: ctx.last_zeros). The
        caller keeps `with torch.cuda.stream(scorer.stream)` around its loop."""
        if self._host_exchange:
            return self.ctx.combine_fast(self._h_ptr, self._score_host(fp))
        total_len = self._enqueue(fp, self.d_part)
        self._all_reduce(self.d_part, dist.ReduceOp.SUM)
        self._fetch()
        return self.ctx.combine_fast(self._h_ptr, total_len)

    def calc_prob_batch(self, path_sets):
        """Several path sets whose values are only compared afterwards: evaluations
        enqueued back to back, ONE all-reduce over all their partials, one device->host copy."""
        k = self.d_part.numel()
        buf = torch.zeros(len(path_sets) * k, dtype=torch.float64, device="cuda")
        with torch.cuda.stream(self.stream):
            tls = [self._enqueue(paths, buf[i * k:(i + 1) * k]) for i, paths in enumerate(path_sets)]
            self._all_reduce(buf, dist.ReduceOp.SUM)
            part = buf.cpu().numpy().reshape(len(path_sets), k)
        out = []
        for i, tl in enumerate(tls):
            prob, zeros = self.ctx.combine_partials(part[i], tl)
            out.append((prob, zeros, tl))
        return out
