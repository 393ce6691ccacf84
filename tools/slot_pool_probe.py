"""Where the continuous-batching leg's wall time goes (bench.py `continuous_batching_queue`): host time per phase of SlotPool.run and, when
run under `rocprofv3 --kernel-trace --stats`, the GPU time per kernel family.    python tools/slot_pool_probe.py [--poll N]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chattts_amd import engine as E, weights as W  # noqa: E402
from chattts_amd.serving import SlotPool  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poll", type=int, default=None)
    ap.add_argument("--nq", type=int, default=4)
    ap.add_argument("--no-codec", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sds = W.synthetic_all()
    gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
    codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev, gemm="f16")
    wl = bench.shard_workload(64, 1, 0, 128, 512)
    ids_t, mask_t, tm_t, stop = torch.from_numpy(wl["ids"]), torch.from_numpy(wl["mask"]), torch.from_numpy(wl["tmask"]), wl["stop"]
    if a.poll:
        SlotPool.POLL = a.poll
    pool = SlotPool(gpt, slots=64, cap=48 + 512 + 2 + 2 * SlotPool.POLL, hid_cap=520, manual_seed=42)
    acc = {"admit_s": 0.0, "admit_n": 0}
    orig_admit = pool._admit

    def timed_admit():
        n0, t0 = pool.admissions, time.perf_counter()
        orig_admit()
        if pool.admissions != n0:
            acc["admit_s"] += time.perf_counter() - t0
            acc["admit_n"] += 1
    pool._admit = timed_admit

    def pool_pass():
        for k in range(a.nq):
            for b in range(ids_t.shape[0]):
                m = mask_t[b].bool()
                pool.submit((k, b), ids_t[b][m], tm_t[b][m], max_new_token=int(stop[b]) + 1, stop_at=int(stop[b]))
        done, pend, n_tok = [], [], 0
        for rid, ids_r, hid_r in pool.run():
            n_tok += int(ids_r.shape[0])
            done.append(hid_r)
            if len(done) == 64 and not a.no_codec:
                pend.append(codec.decode_to_wavs_async(done))
                done = []
        if done and not a.no_codec:
            pend.append(codec.decode_to_wavs_async(done))
        [p_.result() for p_ in pend]
        return n_tok

    pool_pass()
    torch.cuda.synchronize(dev)
    acc.update(admit_s=0.0, admit_n=0)
    s0, a0, t0 = pool.steps, pool.admissions, time.perf_counter()
    n_tok = pool_pass()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    steps = pool.steps - s0
    aud = bench.audio_seconds(stop) * a.nq
    print(f"POLL {SlotPool.POLL}: wall {1e3 * dt:.1f} ms, {steps} decode steps ({1e3 * dt / steps:.4f} ms wall per step), {pool.admissions - a0} admissions, "
          f"host time inside _admit {1e3 * acc['admit_s']:.1f} ms ({1e3 * acc['admit_s'] / max(1, acc['admit_n']):.2f} ms each), tokens {n_tok} "
          f"({n_tok / steps:.2f} per step), {aud / dt:.1f} audio-s/s" + (" (no codec)" if a.no_codec else ""))


if __name__ == "__main__":
    main()
