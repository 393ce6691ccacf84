"""Data-parallel sharding of utterance batches.

Utterances never interact (attention is per sequence, sampling per (b,k) row, DVAE/Vocos per
sequence), so the only collective on the path is ONE broadcast of the packed weights at load
(RCCL over xGMI when the process group backend is "nccl"; gloo in the CPU tests).  The reference has
no data-parallel mode at all; its only distributed code is the optional vLLM fork
(/root/reference/ChatTTS/model/velocity/worker.py:207-238), which is tensor-parallel and out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: sizes differ by at most one, earlier ranks take the extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


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
