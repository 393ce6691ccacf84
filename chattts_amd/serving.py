"""Continuous batching over a pool of utterance slots.

The reference's default path generates one fixed batch to completion (`GPT.generate`, gpt.py:316-618; rows that
finish early idle until the last one is done, :592); only its optional vLLM fork schedules requests continuously
(/root/reference/ChatTTS/model/velocity/scheduler.py:130-293, block_manager.py:73-296).  `SlotPool` is the
MI355X-native equivalent for this engine: a fixed pool of S utterance slots with a dense KV cache per slot
(288 GB of HBM make paging unnecessary: 64 slots x 2560 positions x 61 KB = 10 GB), ONE captured decode graph
for the whole session, and device-side state that makes admission / retirement free of re-capture: the `finish` flags
(a free or retired slot looks finished; the first kernel of every decode step ranks the unfinished slots and computes exactly
those -- device-side compaction, include/chattts_amd.h) and `prompt_len` (where each slot's generated part starts).  Admission
is: write the slot's state, clear its flag.  Newly admitted requests are prefilled as a group straight into their slots' KV cache.

Parity contract: a request produces exactly the tokens `GptEngine.generate` produces for it alone with
`row_offset = 4*slot, total_rows = 4*S` (the Exp(1) draw of a sampling row is the pool row's), because nothing
in the step mixes utterances.  With the host generator, seeded sampling only (`manual_seed`: the reference re-seeds its CPU
generator every step, so the draw is one constant tensor for the whole session); `rng="device"` serves unseeded sampling too.
"""
from __future__ import annotations

import ctypes as C
from collections import deque
from dataclasses import dataclass
from typing import Deque, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._sync import wait_event, wait_stream
from .config import GPT
from .engine import GptEngine, gen_logits, plan_from_processors
from .rng import ExpDraws, penalty_table


@dataclass
class _Req:
    rid: object
    ids: torch.Tensor        # [T, 4] int64
    tmask: torch.Tensor      # [T] bool
    max_new: int
    stop_at: int             # -1: none (benchmark hook, see engine.generate)


class SlotPool:
    POLL = 8   # decode steps between two looks at the finish flags (= admission / retirement granularity)

    def __init__(self, engine: GptEngine, slots: int = 64, cap: int = 1536, hid_cap: int = 1024, *, temperature=(0.3,) * 4,
                 top_P: Optional[float] = 0.7, top_K: Optional[int] = 20, repetition_penalty: float = 1.05, manual_seed: int = 42,
                 min_new_token: int = 0, eos_token: int = GPT.n_audio - 1, rng: str = "host", rng_seed: Optional[int] = None):
        """`rng="device"`: the Exp(1) draws come from the sampling kernel's own generator (engine.generate's `rng`), which is what
        makes the reference's DEFAULT `manual_seed=None` servable here: a fresh draw per (admission, request step, pool row) without any
        per-step host work -- every admission gets its own number as the fourth word of the generator's counter, so a request never
        replays the stream of the slot's previous occupant (`nonce_of[rid]`; `generate(rng_nonce=...)` reproduces it in isolation).  The host stream (`rng="host"`) needs `manual_seed` (one constant tensor per session)."""
        if rng not in ("host", "device"):
            raise ValueError("rng must be 'host' or 'device'")
        if manual_seed is None and rng != "device":
            raise NotImplementedError("SlotPool with the host generator needs manual_seed (one constant Exp(1) draw per session); "
                                      "use rng='device' for unseeded sampling")
        if cap > engine.max_pos:
            raise ValueError("slot capacity exceeds max_position_embeddings")
        self.eng, self.S, self.cap, self.hid_cap = engine, slots, cap, hid_cap
        self.lib = engine.lib
        dev = self.dev = engine.device
        warpers, procs = gen_logits(GPT.n_audio - 1, top_P, top_K, repetition_penalty)
        plan = plan_from_processors((*procs, *warpers))
        h = C.c_void_p()
        _lib.check(self.lib.ctts_gpt_create(C.byref(h), C.byref(engine._w)), "ctts_gpt_create")
        self.handle = h
        self.st = torch.cuda.Stream(device=dev)
        nvq = GPT.n_vq
        with torch.cuda.stream(self.st):
            self.ids_buf = torch.zeros((slots, cap, nvq), dtype=torch.int64, device=dev)
            self.len = torch.ones((slots,), dtype=torch.int32, device=dev)
            self.kv_start = torch.zeros((slots,), dtype=torch.int32, device=dev)
            # finish flags + end_idx in one padded block: a poll is one shader copy of it into pinned memory (engine.snapshot)
            self._Sp = (slots + 15) // 16 * 16
            self.state_blk = torch.zeros((5 * self._Sp,), dtype=torch.uint8, device=dev)
            self.finish = self.state_blk[:slots]
            self.finish.fill_(1)                                                    # free slots look finished
            self.end_idx = self.state_blk[self._Sp:].view(torch.int32)[:slots]
            self.prompt_len = torch.ones((slots,), dtype=torch.int32, device=dev)
            self.stop_at = torch.full((slots,), -1, dtype=torch.int32, device=dev)
            self.hiddens = torch.empty((slots, hid_cap, GPT.hidden), dtype=torch.float32, device=dev)
            kv_shape = (engine.n_layers, slots, GPT.n_heads, cap, GPT.head_dim)
            self.kcache = torch.zeros(kv_shape, dtype=engine.wdt, device=dev)
            self.vcache = torch.zeros(kv_shape, dtype=engine.wdt, device=dev)
            self.n_active = torch.zeros((1,), dtype=torch.int32, device=dev)   # written by the step's first kernel
            self.device_rng, self.rng_per_step = rng == "device", manual_seed is None
            if self.device_rng:
                self.q = torch.zeros((1,), dtype=torch.float32, device=dev)
                seed = int(rng_seed) if rng_seed is not None else (int(manual_seed) if manual_seed is not None
                                                                   else int(torch.randint(0, 2 ** 62, (1,)).item()))
                self.rng_seed = torch.tensor([seed], dtype=torch.int64, device=dev)
                # unseeded (manual_seed=None): a request's step index restarts at 0, so the Philox counter (token group, pool row, step)
                # alone would replay the previous occupant's stream in the same slot.  A per-slot admission number is the fourth counter
                # word (ctts_gen_state.rng_nonce): every (admission, step, row, token) draws fresh.  Seeded pools keep the constant word --
                # the reference re-seeds at every step there, the same draw for every request IS its semantics.
                self.rng_nonce = torch.zeros((slots,), dtype=torch.int32, device=dev) if manual_seed is None else None
            else:
                self.q = ExpDraws(slots * nvq, GPT.n_audio, manual_seed).step(0).to(dev).reshape(1, slots * nvq, GPT.n_audio).contiguous()
                self.rng_seed = None
                self.rng_nonce = None
            self.temp = torch.tensor(list(temperature), dtype=torch.float32, device=dev)
            ptab = penalty_table(plan.penalty)
            self.ptab = None if ptab is None else ptab.to(dev)
            ws_bytes = self.lib.ctts_gpt_workspace_bytes(slots, 1)
            self.ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        self.plan, self.min_new, self.eos = plan, int(min_new_token), int(eos_token)
        self.dec = self._state(B=slots, T=1, workspace=self.ws, row_map=None, n_active=self.n_active)
        self.st.synchronize()
        _lib.check(self.lib.ctts_gpt_graph_build(self.handle, C.byref(self.dec), self.st.cuda_stream), "ctts_gpt_graph_build")
        self.free: List[int] = list(range(slots))
        self.active: dict = {}                 # slot -> (_Req, Tg, first snapshot sequence number that reflects this request)
        self.queue: Deque[_Req] = deque()
        self.steps = 0
        self.admissions = 0                   # prefill groups so far (bench.py reports it)
        self._keep = None
        self.slot_of: dict = {}               # request id -> slot it ran in (parity tests / tracing)
        self.nonce_of: dict = {}              # request id -> its admission number (device generator, unseeded: ctts_gen_state.rng_nonce)

    def close(self):
        if getattr(self, "handle", None):
            self.st.synchronize()
            self.lib.ctts_gpt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _state(self, B, T, workspace, row_map, n_active) -> _lib.GenState:
        s = _lib.GenState()
        s.B, s.T, s.max_new = B, T, self.cap - T
        s.ids_buf, s.len, s.kv_start = self.ids_buf.data_ptr(), self.len.data_ptr(), self.kv_start.data_ptr()
        s.finish, s.end_idx, s.hiddens = self.finish.data_ptr(), self.end_idx.data_ptr(), self.hiddens.data_ptr()
        s.kcache, s.vcache, s.q, s.nq = self.kcache.data_ptr(), self.vcache.data_ptr(), self.q.data_ptr(), 1
        s.temperature, s.pow_table = self.temp.data_ptr(), _lib.ptr(self.ptab)
        p = self.plan
        s.top_p_thr = float(np.float32(1.0 - p.top_p)) if p.top_p is not None else 0.0
        s.use_top_p, s.top_k, s.use_top_k = int(p.top_p is not None), int(p.top_k or 0), int(p.top_k is not None)
        s.min_new, s.eos, s.row_offset = self.min_new, self.eos, 0
        s.stop_at = self.stop_at.data_ptr()
        s.workspace, s.workspace_bytes = workspace.data_ptr(), workspace.numel()
        s.row_map, s.n_active = _lib.ptr(row_map), _lib.ptr(n_active)
        s.cap, s.hid_cap, s.kv_batch, s.q_batch = self.cap, self.hid_cap, self.S, self.S
        s.prompt_len = self.prompt_len.data_ptr()
        s.infer_text = 0
        s.rng_device, s.rng_per_step, s.rng_seed = int(self.device_rng), int(self.rng_per_step), _lib.ptr(self.rng_seed)
        s.rng_nonce = _lib.ptr(getattr(self, "rng_nonce", None))
        return s

    # -- request intake ---------------------------------------------------------------------------------------
    def submit(self, rid, input_ids, text_mask=None, max_new_token: int = 512, stop_at: int = -1) -> None:
        ids = torch.as_tensor(input_ids).to(torch.int64)
        assert ids.dim() == 2 and ids.shape[1] == GPT.n_vq
        tm = torch.ones(ids.shape[0], dtype=torch.bool) if text_mask is None else torch.as_tensor(text_mask).bool()
        # 2 * POLL positions of slack: a request that ends by max_new_token (no EOS) is retired by the HOST, up to two chunks late
        if ids.shape[0] + max_new_token + 1 + 2 * self.POLL > self.cap or max_new_token > self.hid_cap:
            raise ValueError("request does not fit a slot (prompt + max_new_token + 2 * POLL vs cap, max_new_token vs hid_cap)")
        self.queue.append(_Req(rid, ids, tm, int(max_new_token), int(stop_at)))

    def _admit(self) -> None:
        # A group is left-padded to its longest prompt Tg, and every member then needs Tg + max_new_token + 1 <= cap (not just
        # its own prompt length, which is all submit() can check): take queued requests in FIFO order while that holds for the
        # whole group; the first one that does not fit waits for the next round (alone it always fits).  Nothing is popped
        # from the queue / the free list before the group is known to be valid.
        n, Tg, need = 0, 0, 0
        for r in list(self.queue)[: len(self.free)]:
            t_new = max(Tg, int(r.ids.shape[0]))
            need_new = max(need, r.max_new)
            if n > 0 and t_new + need_new + 1 + 2 * self.POLL > self.cap:
                break
            n, Tg, need = n + 1, t_new, need_new
        if n == 0:
            return
        assert Tg + need + 1 + 2 * self.POLL <= self.cap
        reqs = [self.queue.popleft() for _ in range(n)]
        self.admissions += 1
        slots = [self.free.pop(0) for _ in range(n)]              # lowest free slots first (deterministic)
        ids = torch.zeros((n, Tg, GPT.n_vq), dtype=torch.int64)
        mask = torch.zeros((n, Tg), dtype=torch.bool)
        tmask = torch.zeros((n, Tg), dtype=torch.bool)
        for i, r in enumerate(reqs):                               # left padding, like Tokenizer.encode (tokenizer.py:73-110)
            t = int(r.ids.shape[0])
            ids[i, Tg - t:], mask[i, Tg - t:], tmask[i, Tg - t:] = r.ids, True, r.tmask
        dev = self.dev
        with torch.cuda.stream(self.st):
            sl = torch.tensor(slots, dtype=torch.long, device=dev)
            emb = self.eng.embed_prompt(ids, tmask)
            self.ids_buf[sl, :Tg] = ids.to(dev)
            self.len[sl] = Tg
            self.prompt_len[sl] = Tg
            self.kv_start[sl] = (Tg - mask.sum(1)).to(torch.int32).to(dev)
            self.finish[sl] = 0
            self.end_idx[sl] = 0
            self.stop_at[sl] = torch.tensor([r.stop_at for r in reqs], dtype=torch.int32, device=dev)
            if getattr(self, "rng_nonce", None) is not None:
                self._admit_no = getattr(self, "_admit_no", 0) + n
                # globally unique admission numbers (never the constant word of a plain generate() call)
                self.rng_nonce[sl] = torch.arange(self._admit_no - n + 1, self._admit_no + 1, dtype=torch.int32, device=dev)
                for i, r in enumerate(reqs):
                    self.nonce_of[r.rid] = self._admit_no - n + 1 + i
            rmap = sl.to(torch.int32)
            ws = torch.empty((self.lib.ctts_gpt_workspace_bytes(n, Tg),), dtype=torch.uint8, device=dev)
            pre = self._state(B=n, T=Tg, workspace=ws, row_map=rmap, n_active=None)
            _lib.check(self.lib.ctts_gpt_prefill(self.handle, C.byref(pre), emb.data_ptr(), self.st.cuda_stream), "ctts_gpt_prefill")
            self._keep = (ws, emb, rmap, sl)  # stream-ordered: stay alive until the next poll's sync, no extra sync here
        since = getattr(self, "_snap_seq", 0)      # snapshots enqueued before this admission still show the previous occupant
        for s_, r in zip(slots, reqs):
            self.active[s_] = (r, Tg, since)
            self.slot_of[r.rid] = s_

    # -- main loop --------------------------------------------------------------------------------------------
    def _snapshot(self):
        """stream-ordered shader copy of the finish flags + end_idx into one of two pinned blocks, and the event behind it"""
        if not hasattr(self, "_snaps"):
            self._snaps = [(torch.empty((5 * self._Sp,), dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            self._snap_seq = 0
        blk, ev = self._snaps[self._snap_seq % 2]
        _lib.check(self.lib.ctts_copy_bytes(blk.data_ptr(), self.state_blk.data_ptr(), blk.numel(), self.st.cuda_stream), "ctts_copy_bytes")
        ev.record(self.st)
        self._snap_seq += 1
        return self._snap_seq - 1, blk, ev

    def run(self) -> Iterator[Tuple[object, torch.Tensor, torch.Tensor]]:
        """Yields (request id, ids [n,4] int64, hiddens [n,768] float32) as requests complete, admitting queued
        requests into freed slots between decode chunks.  ONE chunk runs ahead: the next POLL steps are enqueued before the host
        looks at the previous chunk's flags, so the device never waits for the poll, the admission prefill or the result copies
        (a freed slot idles for at most two chunks instead of one).  A slot's entry only listens to snapshots enqueued after its
        admission -- an older one still shows the previous occupant's flag."""
        pending: Deque = deque()
        ready: Deque = deque()      # (event behind the result copies, results) of the previous poll
        while self.queue or self.active or pending:
            self._admit()
            if self.active:
                _lib.check(self.lib.ctts_gpt_graph_launch(self.handle, self.POLL, self.st.cuda_stream), "ctts_gpt_graph_launch")
                self.steps += self.POLL
                pending.append(self._snapshot())
            while ready:          # the copies were enqueued in front of the chunk launched just now: the device stays busy while we wait
                ev_out, outs = ready.popleft()
                wait_event(ev_out)
                for o in outs:
                    yield o
            if len(pending) < 2 and self.active:
                continue          # keep one chunk running ahead of the snapshot the host is about to read
            if not pending:
                continue
            seq, blk, ev = pending.popleft()
            wait_event(ev)        # polled, not an interrupt wait (chattts_amd/_sync.py)
            fin = blk[: self.S]
            end = blk[self._Sp:].view(torch.int32)[: self.S]
            done = [s for s, (r, _, since) in self.active.items() if seq >= since and (bool(fin[s]) or int(end[s]) >= r.max_new)]
            if not done:
                continue
            outs = []
            with torch.cuda.stream(self.st):
                for s in done:
                    r, Tg, _ = self.active.pop(s)
                    n = min(int(end[s]), r.max_new)
                    outs.append((r.rid, self.ids_buf[s, Tg: Tg + n].clone(), self.hiddens[s, :n].clone()))
                    self.finish[s] = 1      # a request cut at max_new_token stops costing attention bandwidth
                ev_out = torch.cuda.Event()
                ev_out.record(self.st)
            ready.append((ev_out, outs))    # handed out after the next chunk has been enqueued; the slots are free now (re-admission
            self.free.extend(done)          # writes are stream-ordered behind the copies)
            self.free.sort()
        while ready:
            ev_out, outs = ready.popleft()
            wait_event(ev_out)
            for o in outs:
                yield o
