#!/usr/bin/env python
"""bench.py -- audio-seconds per wall-second of the hot path on N MI355X GPUs (one process per GPU).

A "step" = one pass of the whole hot path over one batch of synthetic input: prefill + autoregressive
decode (hipGraph replay) of a 64-utterance mixed-length batch, then DVAE + Vocos decode of all 64
rows to float32 waveforms (BASELINE.json configs[2], "batch=64 mixed-length utterances ... hipGraph-
captured decode, top-p sampling"; the metric is quoted on batch=64).  Inputs (weights, prompts, the
Exp(1) draws of the seeded CPU generator) are resident in HBM before the timed region starts.

N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` -- the global batch
of 64*N utterances is sharded in contiguous row blocks (weak scaling), weights are broadcast once
from rank 0 over RCCL, there is no collective on the data path.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from chattts_amd import synth, weights as W  # noqa: E402
from chattts_amd.config import GPT, SAMPLE_RATE  # noqa: E402

TAGS = {0: "embed", 1: "qkv_gemm", 2: "rope_append", 3: "attention", 4: "o_proj_gemm", 5: "gate_up_gemm", 6: "down_gemm",
        7: "final_norm", 8: "heads_gemm", 9: "sample"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)


def audio_seconds(lens) -> float:
    return float(sum(256 * (2 * int(t) - 1) for t in lens if t > 0)) / SAMPLE_RATE


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--min-len", type=int, default=128)
    ap.add_argument("--max-len", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-ttfs", action="store_true")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent decode lanes (HIP streams) the batch is cut into")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run: one process per GPU, RCCL process group
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)

    from chattts_amd import dist as D
    from chattts_amd import engine as E

    # ---- weights: rank 0 builds the synthetic checkpoint, everyone else receives it over RCCL ----
    if dist is not None:
        sds = W.synthetic_all() if rank == 0 else None
        sds = D.broadcast_state_dicts(sds, src=0, device=dev, meta=D.weights_meta(GPT.n_layers))
        sds = {n: {k: v.cpu() for k, v in sd.items()} for n, sd in sds.items()}  # the engines repack from host tensors
    else:
        sds = W.synthetic_all()
    gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype=args.dtype)
    codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev)

    # ---- workload: global batch sharded in contiguous row blocks ----
    Bg = args.batch * world
    ids, mask, tmask = synth.make_prompts(Bg, 16, 48, seed=0)
    stop = synth.make_stop_lengths(Bg, args.min_len, args.max_len, seed=0)
    lo, hi = D.shard_bounds(Bg, world, rank)
    ids_t, mask_t, tm_t = torch.from_numpy(ids[lo:hi]), torch.from_numpy(mask[lo:hi]), torch.from_numpy(tmask[lo:hi])
    stop_t = torch.from_numpy(stop[lo:hi])
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    temp = torch.tensor([0.3] * 4)
    max_new = int(stop.max()) + 1
    emb = gpt.embed_prompt(ids_t, tm_t)
    ids_d, mask_d = ids_t.to(dev), mask_t

    def one_pass(use_graph=True, profile_tag=None, decode_audio=True, max_new_override=None, profile_stride=1):
        out = None
        for out in gpt.generate(emb, ids_d, temp, 625, mask_d, max_new_override or max_new, 0, (*procs, *warpers), return_hidden=True,
                                manual_seed=42, use_graph=use_graph, stop_at=stop_t, row_offset=lo * 4, total_rows=Bg * 4,
                                profile_tag=profile_tag, profile_stride=profile_stride, lanes=args.lanes):
            pass
        lens = [int(t.shape[0]) for t in out.ids]
        wav = codec.decode_to_wavs(out.hiddens) if decode_audio else None
        return lens, wav

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        lens, wav = one_pass(use_graph=not args.no_graph)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lens, wav = one_pass(use_graph=not args.no_graph)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert lens == stop[lo:hi].tolist(), "forced lengths not honoured"
    assert wav is not None and bool(torch.isfinite(wav).all())
    total_audio = audio_seconds(stop) * args.steps  # all ranks, all steps
    value = total_audio / dt
    gpt_steps = gpt.last_stats.get("steps", 0)

    result = {
        "metric": "audio seconds/sec (RTF), batch=64 per GPU", "value": round(value, 2), "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "C3: batch=64/GPU mixed-length (prompts 16-48 tok, outputs U{%d..%d} tok), top-p .7/top-k 20/rep 1.05/"
                               "temp .3, manual_seed 42, hipGraph decode + DVAE + Vocos" % (args.min_len, args.max_len),
                   "global_batch": Bg, "decode_steps_per_pass": gpt_steps, "parallelism": f"dp{world}", "lanes_per_gpu": args.lanes,
                   "tokens_per_pass": int(stop.sum()), "audio_s_per_pass": round(audio_seconds(stop), 2)},
    }

    # ---- roofline of the dominant decode kernel: HIP start/stop events of sampled launches (hipExtLaunchKernel, on the launch stream),
    #      in an eager pass of the SAME workload (every 5th launch of the tag over all decode steps) ----
    if rank == 0 and not args.no_roofline:
        per_tag = {}
        for tag in (1, 3, 4, 5, 6, 8, 9):
            calls = 1 if tag in (8, 9) else GPT.n_layers
            stride = 1 if calls == 1 else 5
            one_pass(use_graph=False, profile_tag=tag, profile_stride=stride, decode_audio=False)
            per_tag[tag] = gpt.last_stats.get("profile", (0, 0.0))
        calls_per_step = {t: (1 if t in (8, 9) else GPT.n_layers) for t in per_tag}
        step_ms = {TAGS[t]: round(per_tag[t][1] / max(1, per_tag[t][0]) * calls_per_step[t], 4) for t in per_tag}
        dom = max(per_tag, key=lambda t: per_tag[t][1] / max(1, per_tag[t][0]) * calls_per_step[t])
        n, tot = per_tag[dom]
        avg_ms = tot / max(1, n)
        es = 2 if args.dtype == "bf16" else 4
        B = hi - lo
        valid_prompt = mask[lo:hi].sum(1).astype(np.int64)
        if dom == 3:
            # SURVEY 8d per-unit figure: KV read 2*768*s bytes per visible key per LIVE row per layer, + q read / out write.
            # At decode step i row b sees valid_prompt[b] + i keys and is live while i <= stop[b] (after its EOS the
            # engine drops it from the step -- the reference would keep reading its KV, but no output depends on it, so
            # those bytes are not counted as useful work).  Launches are sampled uniformly over decode steps 1..steps-1.
            st_ = stop[lo:hi].astype(np.int64)
            ctx = [((valid_prompt + i) * (st_ >= i)).sum() for i in range(1, gpt_steps)]
            live = [int((st_ >= i).sum()) for i in range(1, gpt_steps)]
            alg = float(np.mean(ctx)) * 2 * 768 * es + float(np.mean(live)) * 768 * (4 + es)
        else:
            wbytes = {1: 3 * 768 * 768 * es, 4: 768 * 768 * es, 5: 2 * 3072 * 768 * es, 6: 768 * 3072 * es, 8: 2504 * 768 * 4}.get(dom, 0)
            act = {1: B * (768 * es + 2304 * 4), 4: B * 768 * (es + 8 + es), 5: B * (768 + 3072) * es, 6: B * (3072 * es + 768 * (8 + es)),
                   8: B * (768 + 2504) * 4, 9: B * 4 * 626 * 8}.get(dom, 0)
            alg = float(wbytes + act)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC counters: collected by `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
        # passes of THIS command (tools/gpu_round.sh pmc -> profiles/pmc_traffic.json, gfx950 FETCH x2 correction applied)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                traffic = json.load(fh).get(TAGS[dom], {}).get("hbm_bytes_per_launch")
        except OSError:
            pass
        result["roofline"] = {"kernel": TAGS[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_us": round(avg_ms * 1e3, 2),
                              "launches_timed": n, "alg_bytes_per_launch": int(alg)}
        result["decode_kernel_ms_per_step"] = step_ms

    # ---- time to first sample: stream=True with the reference's yield schedule (first audio after 3 x 24 tokens) ----
    if rank == 0 and not args.no_ttfs:
        from chattts_amd.core import Chat, InferCodeParams
        chat = Chat()
        chat.gpt, chat.codec = gpt, codec
        params = InferCodeParams(max_new_token=max_new, manual_seed=42, show_tqdm=False)
        ttfs = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for chunk in chat.infer_ids_stream(ids_t, mask_t, tm_t, params, stop_at=stop_t, row_offset=lo * 4, total_rows=Bg * 4):
                ttfs.append(time.perf_counter() - t1)   # chunk is a host numpy array: audio is on the host here
                break
        result["ttfs_ms_p50"] = round(1000.0 * float(np.median(ttfs)), 2)

    # ---- same-box CPU baseline: the numpy port of the reference path, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sds, ids[lo:hi], mask[lo:hi], tmask[lo:hi])

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(sds, ids, mask, tmask, n_steps: int = 12, codec_rows: int = 2, codec_T: int = 96):
    """oracle/ (numpy float32 port of GPT.generate + DVAE + Vocos) timed on this box's host cores:
    batch-64 prefill + `n_steps` decode steps, plus DVAE+Vocos on a [codec_rows, codec_T] slice;
    rate = tokens / (gpt_time + codec_time_per_token * tokens) / 46.875."""
    from chattts_amd import rng
    from oracle import codec_np, generate_np, llama_np

    cores = os.cpu_count() or 1
    llama = llama_np.LlamaWeights({k: v.float().cpu().numpy() for k, v in sds["gpt"].items()})
    esd = {k: v.float().cpu().numpy() for k, v in sds["embed"].items()}
    heads = generate_np.fold_heads(esd)
    emb = generate_np.embed_prompt(esd, ids, tmask)
    B = ids.shape[0]
    draws = rng.ExpDraws(B * 4, 626, 42)
    t0 = time.perf_counter()
    res = generate_np.generate(llama, esd, heads, emb, ids, mask, temperature=np.array([0.3] * 4, np.float32),
                               draw_q=lambda i: draws.step(i).numpy(), pow_table=rng.penalty_table(1.05).numpy(),
                               max_new_token=n_steps, min_new_token=n_steps)
    t_gpt = time.perf_counter() - t0
    tokens = B * n_steps
    dsd = {k: v.float().cpu().numpy() for k, v in sds["decoder"].items()}
    vsd = {k: v.float().cpu().numpy() for k, v in sds["vocos"].items()}
    hid = np.random.RandomState(0).standard_normal((codec_rows, codec_T, 768)).astype(np.float32)
    t0 = time.perf_counter()
    codec_np.vocos_decode(vsd, codec_np.dvae_decode(dsd, hid))
    t_codec_per_tok = (time.perf_counter() - t0) / (codec_rows * codec_T)
    wall = t_gpt + t_codec_per_tok * tokens
    return {"value": round(tokens / 46.875 / wall, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"numpy oracle: B={B} prefill + {n_steps} decode steps ({t_gpt:.1f}s) + DVAE/Vocos on {codec_rows}x{codec_T} tokens "
                      f"({t_codec_per_tok * 1e3:.2f} ms/token); note: short contexts (<= {ids.shape[1] + n_steps}) favour the CPU"}


if __name__ == "__main__":
    main()
