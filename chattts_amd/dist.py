"""Data-parallel sharding of utterance batches.

Utterances never interact (attention is per sequence, sampling per (b,k) row, DVAE/Vocos per
sequence), so the only collective on the path is ONE broadcast of the packed weights at load
(RCCL over xGMI when the process group backend is "nccl"; gloo in the CPU tests).  The reference has
no data-parallel mode at all; its only distributed code is the optional vLLM fork
(/root/reference/ChatTTS/model/velocity/worker.py:207-238), which is tensor-parallel and out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: sizes differ by at most one, earlier ranks take the extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def deal_shards(lengths: Sequence[int], world: int, policy: str = "snake") -> List[List[int]]:
    """Which utterances each rank generates ("sorted by expected length to balance").  The expected work of an utterance
    follows its prompt length (the text decides how much speech there is); a rank's wall time is set by its longest utterance and by how
    many rows stay alive how long, so the ranks should get EQUAL shares of long and short utterances:
      "snake"  (default) utterances sorted by length, longest first (stable: ties keep the caller's order), dealt 0..W-1, W-1..0, ...;
      "blocks" the caller's order cut into contiguous blocks (`shard_bounds`; the layout of rounds 1-5, no permutation);
      "sorted_blocks" contiguous blocks of the sorted order (homogeneous shards: least padding, but the longest utterances share a rank).
    Every shard lists its utterances in ASCENDING caller index, so a world of one is the unsharded call, row for row."""
    n = len(lengths)
    if policy == "blocks":
        return [list(range(*shard_bounds(n, world, r))) for r in range(world)]
    order = sorted(range(n), key=lambda i: (-int(lengths[i]), i))
    if policy == "sorted_blocks":
        return [sorted(order[slice(*shard_bounds(n, world, r))]) for r in range(world)]
    if policy != "snake":
        raise ValueError("policy must be 'snake', 'blocks' or 'sorted_blocks'")
    shards: List[List[int]] = [[] for _ in range(world)]
    for j, i in enumerate(order):
        rnd, pos = divmod(j, world)
        shards[pos if rnd % 2 == 0 else world - 1 - pos].append(i)
    return [sorted(s) for s in shards]


def _world_rank(group=None) -> Tuple[int, int, Optional[object]]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group), dist
    return 1, 0, None


def infer_sharded(chat, input_ids: torch.Tensor, attention_mask: torch.Tensor, text_mask: torch.Tensor, params=None, *, policy: str = "snake",
                  gather: bool = True, dst: int = 0, group=None, stop_at: Optional[torch.Tensor] = None, use_decoder: bool = True,
                  return_ids: bool = False, **kw):
    """`Chat.infer_ids` (core.py:469-481: one non-stream batch, BEFORE the silence strip) over the ranks of the process group: every rank
    holds the whole tokenised batch (`Tokenizer.encode` is host work and deterministic), generates ITS utterances (`deal_shards`) and decodes
    them; rank `dst` gets the [B, 256 (2 Tmax - 1)] float32 batch in the CALLER'S order -- what the unsharded call returns.

    What keeps it equal to the unsharded call (SURVEY 8e caveats): the shard keeps the global padded prompt geometry (same left padding,
    same positions); `row_ids` hands every utterance its GLOBAL index, so its Exp(1) draws are its rows of the full-batch CPU draw and the
    rows >= 625 repetition-penalty quirk (processors.py:24-27) follows the caller's numbering (GptEngine.generate, ctts_gen_state.row_base);
    the acoustic decoder pads every shard to the GLOBAL longest utterance (`pad_to`: the reference decodes a shorter row's tail from zero
    hidden states, core.py:525-533, and returns that tail) -- ONE 8-byte all-reduce(max) between generation and decoding, the only exchange
    on the data path besides the result gather itself.  No collective touches a token or a sample while it is being computed.

    gather=False: no result collective; every rank returns (its utterance indices, its rows) -- a server whose ranks answer their own
    requests, and what bench.py times.  The seeded step-0 EOS rule (gpt.py:527-570: the call yields nothing) is kept for the whole batch:
    if any rank's generation yields nothing, the result is empty everywhere.
    `chat`: anything with `infer_code(ids, mask, text_mask, params, stream=False, return_hidden=..., **kw)` and
    `decode_to_wavs(rows, use_decoder, pad_to=...)` (chattts_amd.core.Chat; the CPU tests plug a stand-in)."""
    world, rank, dist = _world_rank(group)
    B = int(input_ids.shape[0])
    nvq = int(input_ids.shape[2])
    lengths = attention_mask.to(torch.int64).sum(1).tolist()
    mine = deal_shards(lengths, world, policy)[rank]
    sel = torch.tensor(mine, dtype=torch.long)
    out = None
    if mine:
        skw = dict(kw)
        if world > 1 or mine != list(range(B)):
            skw.update(row_ids=sel, total_rows=B * nvq)
        if stop_at is not None:
            skw["stop_at"] = stop_at[sel]
        for out in chat.infer_code(input_ids[sel], attention_mask[sel], text_mask[sel], params, stream=False, return_hidden=use_decoder, **skw):
            pass
    rows = [] if out is None else list(out.hiddens if use_decoder else out.ids)
    t_local = max((int(r.shape[0]) for r in rows), default=0)
    empty = 1 if (mine and out is None) else 0
    t_max, any_empty = t_local, empty
    if dist is not None and world > 1:
        dev = input_ids.device if dist.get_backend(group) == "nccl" and input_ids.is_cuda else (
            torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        t = torch.tensor([t_local, empty], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)      # the one exchange before decoding: global Tmax (+ the step-0 rule)
        t_max, any_empty = int(t[0]), int(t[1])
    wav = None
    if mine and not any_empty and t_max > 0:
        wav = chat.decode_to_wavs(rows, use_decoder, pad_to=t_max)
    ids_rows = None if (out is None or not return_ids) else [np.asarray(t.cpu()) for t in out.ids]
    if not gather:
        return (mine, wav, ids_rows) if return_ids else (mine, wav)
    parts = [(mine, wav, ids_rows)]
    if dist is not None and world > 1:
        box = [None] * world if rank == dst else None
        dist.gather_object((mine, wav, ids_rows), box, dst=dst, group=group)
        if rank != dst:
            return None
        parts = box
    if any_empty or t_max == 0:
        return (np.zeros((0,), np.float32), []) if return_ids else np.zeros((0,), np.float32)
    width = next(w.shape[1] for _, w, _ in parts if w is not None)
    full = np.zeros((B, width), np.float32)
    ids_all = [None] * B
    for idx, w, ir in parts:
        for j, b in enumerate(idx):
            full[b] = w[j]
            if ir is not None:
                ids_all[b] = ir[j]
    return (full, ids_all) if return_ids else full


class CapiComm:
    """An RCCL communicator made through the library's own C ABI (`ctts_rccl_unique_id` / `ctts_rccl_comm_create`,
    include/chattts_amd.h) -- what a host without torch would use for the one collective of the path.  The 128-byte unique id travels
    from `src` to the other ranks by the host's own means: here `exchange(id_bytes_or_None) -> id_bytes` (default: torch.distributed's
    object broadcast on whatever backend the process group has; world 1 needs none)."""

    def __init__(self, world: int, rank: int, src: int = 0, exchange=None):
        import ctypes as C

        from . import _lib
        self.lib = _lib.lib()
        self.world, self.rank = int(world), int(rank)
        idb = (C.c_char * 128)()
        if rank == src:
            _lib.check(self.lib.ctts_rccl_unique_id(C.cast(idb, C.c_void_p)), "ctts_rccl_unique_id")
        raw = bytes(idb)
        if world > 1:
            if exchange is None:
                import torch.distributed as dist

                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=src)
                    return box[0]
            raw = exchange(raw if rank == src else None)
        idb2 = (C.c_char * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        _lib.check(self.lib.ctts_rccl_comm_create(C.byref(h), self.world, C.cast(idb2, C.c_void_p), self.rank), "ctts_rccl_comm_create")
        self.handle = h

    def broadcast(self, tensors: List[torch.Tensor], root: int = 0, stream=None) -> None:
        """in place, one ncclBroadcast per tensor (contiguous device tensors), on `stream` (default: torch's current stream)"""
        import ctypes as C

        from . import _lib
        ts = [t for t in tensors if t.numel() > 0]
        assert all(t.is_contiguous() and t.is_cuda for t in ts)
        if not ts:
            return
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        sizes = (C.c_size_t * len(ts))(*[t.numel() * t.element_size() for t in ts])
        st = (stream or torch.cuda.current_stream(ts[0].device)).cuda_stream
        _lib.check(self.lib.ctts_broadcast_weights(ptrs, sizes, len(ts), self.handle, int(root), st), "ctts_broadcast_weights")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ctts_rccl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def broadcast_state_dicts(sds: Dict[str, Dict[str, torch.Tensor]] | None, src: int = 0, device=None,
                          meta: Dict[str, Dict[str, Tuple[tuple, torch.dtype]]] | None = None, comm: "CapiComm | None" = None):
    """Rank `src` holds the real state dicts; every other rank passes `sds=None` and `meta` (key ->
    (shape, dtype), cheap to build from the config).  All tensors of a dict are packed into one flat
    buffer per dtype so the broadcast is a few large messages (ring broadcast over xGMI is per-link
    bound, ~153 GB/s: 0.9 GB of f32 weights ~ 6 ms) instead of ~500 small ones.  `comm`: broadcast through the
    library's C ABI (`ctts_broadcast_weights`) on that communicator instead of torch.distributed's collective."""
    import torch.distributed as dist

    rank = dist.get_rank()
    if rank == src:
        meta = {n: {k: (tuple(v.shape), v.dtype) for k, v in sd.items()} for n, sd in sds.items()}
    assert meta is not None
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for name in sorted(meta):
        keys = sorted(meta[name])
        by_dtype: Dict[torch.dtype, List[str]] = {}
        for k in keys:
            by_dtype.setdefault(meta[name][k][1], []).append(k)
        out[name] = {}
        for dt, ks in by_dtype.items():
            numel = sum(int(torch.Size(meta[name][k][0]).numel()) for k in ks)
            flat = torch.empty(numel, dtype=dt, device=device)
            if rank == src:
                off = 0
                for k in ks:
                    n = sds[name][k].numel()
                    flat[off: off + n] = sds[name][k].reshape(-1).to(flat.device)
                    off += n
            if comm is not None:
                comm.broadcast([flat], root=src)
            else:
                dist.broadcast(flat, src=src)
            off = 0
            for k in ks:
                shape = meta[name][k][0]
                n = int(torch.Size(shape).numel())
                out[name][k] = flat[off: off + n].view(shape)
                off += n
    return out


def weights_meta(n_layers: int) -> Dict[str, Dict[str, Tuple[tuple, torch.dtype]]]:
    """Shapes of the four hot-path state dicts without materialising them."""
    from . import weights as W

    # cheap: a 1-layer synthetic GPT gives the per-layer key set; scale to n_layers
    sd1 = W.synthetic_gpt(n_layers=1)
    gpt = {}
    for k, v in sd1.items():
        if k.startswith("layers.0."):
            for i in range(n_layers):
                gpt[k.replace("layers.0.", f"layers.{i}.")] = (tuple(v.shape), v.dtype)
        else:
            gpt[k] = (tuple(v.shape), v.dtype)
    m = lambda sd: {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
    return {"gpt": gpt, "embed": m(W.synthetic_embed()), "decoder": m(W.synthetic_decoder()), "vocos": m(W.synthetic_vocos())}
