"""Reference-generated golden of the BENCH workload itself (BASELINE.json configs[2], C3): 64 mixed-length left-padded
utterances, default sampling at manual_seed 42, per-row forced output lengths U{128..512}.

Run in the build container only (~6 min of reference CPU time):   python -m oracle.make_bench_golden
`--world N` (2, 4, 8): the GLOBAL batch `bench.py --gpus N` deals to its ranks (64 N utterances, the reference has no data-parallel mode: it
runs them as ONE batch) -> tests/golden/bench_c3_w{N}.npz, what every rank compares its shard's rows with.  Measured on 8 vCPU: 1238 /
1362 / 2884 s; N = 8 (BASELINE config C4, 512 utterances) peaks at 37 GB (35 GB of f32 KV cache + one layer's concatenation copy).

The reference has no length-forcing feature; SURVEY.md 8d prescribes a harness-side logits processor that is identical
on both sides: EOS is masked while fewer than N_b tokens exist and forced from then on.  It is appended LAST in the
processor chain (after top-p / top-k), which is where `min_new_token` acts in the reference (gpt.py:494-495) and where
the HIP sampling kernel applies its `stop_at` hook.  Output: tests/golden/bench_c3.npz (lens, ids as int16, sha256 of the
int64 ids) -- what bench.py's f32 parity leg and tests/test_gpu_e2e.py compare with.  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chattts_amd import synth, weights as W  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def bench_workload(batch=64, min_len=128, max_len=512):
    ids, mask, tmask = synth.make_prompts(batch, 16, 48, seed=0)
    stop = synth.make_stop_lengths(batch, min_len, max_len, seed=0)
    return ids, mask, tmask, stop


def ids_digest(rows) -> str:
    h = hashlib.sha256()
    for r in rows:
        h.update(np.ascontiguousarray(r, dtype=np.int64).tobytes())
    return h.hexdigest()


class ForceLength:
    """harness-side processor: rows of utterance b cannot emit EOS before N_b tokens and must emit it from then on"""

    def __init__(self, stop, eos=625):
        self.sa = torch.from_numpy(np.repeat(stop.astype(np.int64), 4))
        self.eos = eos

    def __call__(self, input_ids, scores):
        i = input_ids.shape[1]          # tokens generated so far (history only, gpt.py:466-475)
        mask = i < self.sa
        force = ~mask
        scores[mask, self.eos] = -torch.inf
        keep = scores[force, self.eos].clone()
        scores[force] = -torch.inf
        scores[force, self.eos] = torch.where(torch.isfinite(keep), keep, torch.zeros_like(keep))
        return scores


def main():
    assert ref_harness.available()
    torch.set_num_threads(os.cpu_count())
    world = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 1
    name = "bench_c3.npz" if world == 1 else "bench_c3_w%d.npz" % world
    sds = W.synthetic_all()
    embed, gpt = ref_harness.build_gpt(sds)
    ids, mask, tmask, stop = bench_workload(64 * world)
    print(name, "batch", ids.shape[0], "threads", torch.get_num_threads(), "load average", os.getloadavg(), flush=True)
    t0 = time.time()
    res, emb, _ = ref_harness.run_generate(embed, gpt, ids, mask, tmask, temperature=[0.3] * 4, top_P=0.7, top_K=20,
                                           repetition_penalty=1.05, max_new_token=int(stop.max()) + 1, min_new_token=0,
                                           manual_seed=42, extra_processors=(ForceLength(stop),))
    lens = np.array([r.shape[0] for r in res.ids], dtype=np.int64)
    assert np.array_equal(lens, stop), (lens, stop)
    rows = [r.numpy() for r in res.ids]
    flat = np.concatenate(rows, 0)
    assert flat.max() < 32767
    np.savez_compressed(os.path.join(OUT, name), lens=lens, ids=flat.astype(np.int16),
                        sha256=np.array(ids_digest(rows)), hid0_first=res.hiddens[0][:4].numpy(), hid0_last=res.hiddens[0][-4:].numpy())
    print(name, lens.sum(), "tokens", ids_digest(rows), f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
