"""CPU-side tests: the C-ABI library loads and exports every declared symbol (no compute calls),
host logic of the engine, sharding + weight broadcast over gloo (world_size 2)."""
import ctypes as C
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from chattts_amd import _lib, dist as D, engine as E, rng, synth  # noqa: E402


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "chattts_amd.h")).read()
    declared = set(re.findall(r"\b(ctts_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()  # dlopen + getattr of every symbol
    assert lib.ctts_version() == 1
    assert lib.ctts_gpt_workspace_bytes(64, 48) > 64 * 48 * (768 + 2304 + 768 + 3072) * 4
    assert lib.ctts_codec_workspace_bytes(2, 10) > 0


def test_ctypes_signatures_match_the_header_prototypes():
    """every prototype of include/chattts_amd.h against chattts_amd/_lib.py SIGNATURES: same number of parameters, and each
    parameter of the same class (pointer / int32 / float / size_t = uint64 (one ctypes type on LP64) / int64) -- a prototype edited on one side only fails on CPU"""
    hdr = open(os.path.join(ROOT, "include", "chattts_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t|void|const char\*|int32_t)\s+(ctts_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert {n for n, _ in protos} == set(_lib.SIGNATURES)

    def klass_c(param: str) -> str:
        param = " ".join(param.split())
        if param == "void":
            return ""
        if "*" in param:
            return "ptr"
        for t, k in (("int32_t", "i32"), ("uint64_t", "size"), ("int64_t", "i64"), ("size_t", "size"), ("float", "f32"), ("int", "i32")):
            if re.match(rf"(const )?{t}\b", param):
                return k
        raise AssertionError(f"unclassified parameter {param!r}")

    def klass_py(t) -> str:
        if t in (C.c_void_p, C.c_char_p) or isinstance(t, type) and issubclass(t, (C._Pointer,)):
            return "ptr"
        return {C.c_int32: "i32", C.c_int: "i32", C.c_int64: "i64", C.c_size_t: "size", C.c_float: "f32"}[t]

    for name, params in protos:
        want = [k for k in (klass_c(p) for p in params.split(",")) if k]
        got = [klass_py(t) for t in _lib.SIGNATURES[name][1]]
        assert got == want, (name, got, want)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """every struct that crosses the C ABI: sizeof and the offset of every field as gcc lays out include/chattts_amd.h
    == what chattts_amd/_lib.py tells ctypes (a field added, dropped or reordered on one side only fails here, on CPU)"""
    import subprocess
    pairs = [("ctts_gpt_weights", _lib.GptWeights), ("ctts_gen_state", _lib.GenState), ("ctts_codec_weights", _lib.CodecWeights),
             ("ctts_trunk_weights", _lib.TrunkWeights), ("ctts_dvae_weights", _lib.DvaeWeights)]
    lines = ['#include <stdio.h>', '#include "chattts_amd.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("%zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("%zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = []
    for _, cls in pairs:
        want.append(C.sizeof(cls))
        want += [getattr(cls, fname).offset for fname, *_ in cls._fields_]
    assert got == want


def test_product_path_never_touches_the_oracle_or_the_reference():
    """the oracle is test infrastructure: nothing under chattts_amd/ imports or executes `oracle`, reads tests/golden or /root/reference
    (comments / docstrings may NAME them); bench.py reaches `oracle` only from its cpu_baseline leg"""
    import ast
    pkg = os.path.join(ROOT, "chattts_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        src = open(os.path.join(pkg, fn)).read()
        tree = ast.parse(src)
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                assert not any(a.name.split(".")[0] == "oracle" for a in node.names), fn
            if isinstance(node, ast.ImportFrom):
                assert (node.module or "").split(".")[0] != "oracle", fn
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and not node.value.count("\n"):
                assert "/root/reference" not in node.value or fn == "weights.py" or "core.py" in node.value, (fn, node.value[:80])
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = set()
    for fn_node in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn_node):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                users.add(fn_node.name)
    assert users <= {"cpu_baseline", "main"}, users      # main: only the post-hoc golden comparison helpers, never inside a timed pass


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EngineError):
        E.GptEngine({}, {}, torch.device("cpu"))
    with pytest.raises(_lib.EngineError):
        E.CodecEngine({}, {}, torch.device("cpu"))


def test_plan_from_processors_accepts_reference_objects():
    from transformers.generation import TopKLogitsWarper, TopPLogitsWarper
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    p = E.plan_from_processors((*procs, *warpers))
    assert (p.penalty, p.top_p, p.top_k) == (1.05, 0.7, 20)
    p = E.plan_from_processors([TopPLogitsWarper(0.5, min_tokens_to_keep=3), TopKLogitsWarper(7, min_tokens_to_keep=3)])
    assert (p.penalty, p.top_p, p.top_k) == (None, 0.5, 7)
    assert E.plan_from_processors(E.gen_logits(625, None, None, 1.0)[0]) == E.SamplingPlan()
    with pytest.raises(NotImplementedError):
        E.plan_from_processors([lambda ids, scores: scores])
    with pytest.raises(NotImplementedError):  # wrong order
        E.plan_from_processors([E.TopK(5), E.TopP(0.5)])
    # refine-text mode: warpers are fine, a repetition penalty is rejected (the reference's processor mis-broadcasts there)
    assert E.plan_from_processors(E.gen_logits(21178, 0.7, 20, 1.0)[0], infer_text=True).top_k == 20
    with pytest.raises(NotImplementedError):
        E.plan_from_processors(E.gen_logits(21178, 0.7, 20, 1.2)[1], infer_text=True)


def test_split_bf16_reconstructs_f32():
    w = torch.randn(37, 100) * 3
    p = E.split_bf16(w)
    assert p.shape == (37, 4, 2, 32) and p.dtype == torch.bfloat16 and p.is_contiguous()   # [N][k block][hi|lo][32]
    rec = (p[:, :, 0].float() + p[:, :, 1].float()).reshape(37, 128)
    assert float((rec[:, :100] - w).abs().max() / w.abs().max()) < 2 ** -15
    assert float(rec[:, 100:].abs().max()) == 0.0


def test_rope_row_perm_is_a_permutation_pairing_halves():
    perm = E.rope_row_perm()
    assert sorted(perm.tolist()) == list(range(768))
    t = perm.view(12, 4, 16)
    assert torch.equal(t[:, :, 8:] - t[:, :, :8], torch.full((12, 4, 8), 32))


def test_fp16_plane_packing_matches_the_split_planes_layout():
    """pack_h1p (the single fp16 plane of gemm_h1p_k) uses the fragment order of pack_x3p's planes without the hi | lo axis: on
    bf16-representable values the two agree element for element; values beyond the half range saturate; round trip"""
    from chattts_amd.engine import pack_h1p, pack_x3p, unpack_h1p
    g = torch.Generator().manual_seed(5)
    w = torch.randn(96, 64, generator=g).to(torch.bfloat16).to(torch.float32)      # exactly representable in bf16 and (here) in fp16
    p1, p3 = pack_h1p(w), pack_x3p(w)
    assert p1.dtype == torch.float16 and tuple(p1.shape) == (3, 4, 2, 32, 8) and tuple(p3.shape) == (3, 4, 2, 2, 32, 8)
    assert torch.equal(p1.to(torch.float32), p3[:, :, 0].to(torch.float32)) and not bool(p3[:, :, 1].to(torch.float32).any())
    assert torch.equal(unpack_h1p(p1, 96, 64), w)
    big = torch.tensor([[1e6, -1e6] + [0.0] * 14] * 32)
    assert unpack_h1p(pack_h1p(big), 32, 16)[0, :2].tolist() == [65504.0, -65504.0]


def test_fragment_packing_round_trips_and_matches_the_kernel_offsets():
    """pack_frag / pack_frag32 (engine.py) against the device-side offset functions they must agree with (csrc/common.hpp
    pk_off / pk32_off, restated here): element (r, c) of a [R, C] matrix lands where the decode kernels look for it"""
    from chattts_amd.engine import pack_frag, pack_frag32, unpack_frag, unpack_frag32
    R, Cc = 48, 96
    w = torch.arange(R * Cc, dtype=torch.float32).reshape(R, Cc)

    def pk_off(m, c, kch):      # bf16: [rows/16][C/32][lane = (c%32)/8*16 + row%16][c%8]
        return (((m >> 4) * kch + (c >> 5)) * 64 + (((c & 31) >> 3) << 4) + (m & 15)) * 8 + (c & 7)

    def pk32_off(m, c, kch):    # f32: [rows/16][C/16][lane = (c%16)/4*16 + row%16][c%4]
        return (((m >> 4) * kch + (c >> 4)) * 64 + (((c & 15) >> 2) << 4) + (m & 15)) * 4 + (c & 3)

    p16, p32 = pack_frag(w).flatten(), pack_frag32(w).flatten()
    for m, c in [(0, 0), (1, 0), (0, 1), (15, 31), (16, 32), (17, 9), (47, 95), (33, 70)]:
        assert p16[pk_off(m, c, Cc // 32)] == w[m, c]
        assert p32[pk32_off(m, c, Cc // 16)] == w[m, c]
    assert torch.equal(unpack_frag(pack_frag(w), R, Cc), w) and torch.equal(unpack_frag32(pack_frag32(w), R, Cc), w)


def test_asset_layout_round_trip(tmp_path):
    """the four hot-path safetensors files in the reference's asset layout (config.py:4-11): save -> load is lossless,
    HF's `model.` prefix and the unused embed_tokens (gpt.py:78) are handled"""
    from chattts_amd import weights as W
    from safetensors.torch import save_file
    sds = {"gpt": W.synthetic_gpt(n_layers=1), "embed": {"emb_code.0.weight": torch.randn(626, 8)},
           "decoder": {"coef": torch.rand(1, 100, 1)}, "vocos": {"head.istft.window": torch.hann_window(1024)}}
    W.save_assets(str(tmp_path), sds)
    got = W.load_assets(str(tmp_path), validate=False)      # toy dicts: the layout check is test_asset_validation's subject
    for name in sds:
        assert set(got[name]) == set(sds[name])
        assert all(torch.equal(got[name][k], sds[name][k]) for k in sds[name])
    # a checkpoint saved from LlamaForCausalLM-style naming: keys prefixed with "model." + an embed_tokens table
    pref = {"model." + k: v for k, v in sds["gpt"].items()}
    pref["model.embed_tokens.weight"] = torch.zeros(4, 768)
    save_file(pref, os.path.join(str(tmp_path), "gpt", "model.safetensors"))
    got = W.load_assets(str(tmp_path), validate=False)["gpt"]
    assert set(got) == set(sds["gpt"]) and W.gpt_layer_count(got) == 1


def test_asset_validation_and_gpt_config(tmp_path, weights):
    """first contact with a checkpoint is a diff-style message, not a KeyError in the repacking: `load_assets` checks every state
    dict against SURVEY App. B (`expected_schema`: key set, shapes, dtypes), accepts the reference's `<root>/asset/...` layout
    (config.py:4-11), and reads `asset/gpt/config.json` like `LlamaModel.from_pretrained` (gpt.py:75) -- geometry fields must be
    the ones the kernels are built for, rms_norm_eps / rope_theta / max_position_embeddings come back as run-time parameters"""
    import json
    from chattts_amd import weights as W
    for name in W.ASSET_FILES:      # the synthetic recipe IS the documented layout
        W.validate_state_dict(name, weights[name])
        assert set(W.expected_schema(name)) == set(weights[name])
    small = dict(weights, gpt=W.synthetic_gpt(n_layers=2))
    root = os.path.join(str(tmp_path), "dl")
    W.save_assets(os.path.join(root, "asset"), small)
    with open(os.path.join(root, "asset", "gpt", "config.json"), "w") as fh:
        json.dump({"hidden_size": 768, "intermediate_size": 3072, "num_attention_heads": 12, "num_hidden_layers": 2, "hidden_act": "silu",
                   "rms_norm_eps": 1e-5, "rope_theta": 50000.0, "max_position_embeddings": 2048, "vocab_size": 21178}, fh)
    got = W.load_assets(root)       # the directory that HOLDS asset/
    assert got["gpt_config"] == {"rms_eps": 1e-5, "rope_theta": 50000.0, "max_pos": 2048}
    assert W.gpt_layer_count(got["gpt"]) == 2 and set(got["vocos"]) == set(weights["vocos"])
    # a Vocos file with other key names / a wrong shape / a missing tensor: every problem is named
    bad = {("backbone.convnext." + k[len("backbone.convnext."):].replace("gamma", "layer_scale") if k.startswith("backbone.convnext.0.g") else k): v
           for k, v in weights["vocos"].items()}
    bad["head.out.weight"] = torch.zeros(1026, 256)
    del bad["head.istft.window"]
    with pytest.raises(W.AssetError) as ei:
        W.validate_state_dict("vocos", bad, where="Vocos.safetensors")
    msg = str(ei.value)
    assert "missing (2)" in msg and "backbone.convnext.0.gamma" in msg and "head.istft.window" in msg
    assert "unexpected (1)" in msg and "backbone.convnext.0.layer_scale" in msg
    assert "wrong shape (1)" in msg and "(1026, 256) != expected (1026, 512)" in msg
    W.validate_state_dict("vocos", dict(weights["vocos"], **{"feature_extractor.mel_spec.mel_scale.fb": torch.zeros(513, 100)}))   # ignored extras
    # config.json for another geometry / another layer count
    with pytest.raises(W.AssetError, match="hidden_size = 1024"):
        W.check_gpt_config({"hidden_size": 1024, "num_attention_heads": 16})
    with pytest.raises(W.AssetError, match="holds 2 layers"):
        W.check_gpt_config({"num_hidden_layers": 20}, 2)
    with pytest.raises(W.AssetError, match="rope_scaling"):
        W.check_gpt_config({"rope_scaling": {"rope_type": "llama3", "factor": 8.0}})
    assert W.check_gpt_config(None) == {"rms_eps": 1e-6, "rope_theta": 10000.0, "max_pos": 4096}
    os.remove(os.path.join(root, "asset", "Decoder.safetensors"))
    with pytest.raises(W.AssetError, match="Decoder.safetensors not found"):
        W.load_assets(root)


def test_validate_assets_tool(tmp_path, weights):
    """tools/validate_assets.py: header-only diff of a checkpoint directory against SURVEY App. B + the third-party
    pins.  A synthetic asset directory written with the reference's key names passes; a Vocos file with other names fails with the diff;
    the packages that are absent here (vocos, vector_quantize_pytorch, torchaudio) are reported as such, not as errors."""
    import importlib.util
    import json
    from chattts_amd import weights as W
    spec = importlib.util.spec_from_file_location("validate_assets", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "tools", "validate_assets.py"))
    va = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(va)
    root = os.path.join(str(tmp_path), "dl")
    small = dict(weights, gpt={("model." + k): v for k, v in W.synthetic_gpt(n_layers=2).items()})     # HF-style prefix + embed_tokens
    small["gpt"]["model.embed_tokens.weight"] = torch.zeros(4, 768)
    W.save_assets(os.path.join(root, "asset"), small)
    with open(os.path.join(root, "asset", "gpt", "config.json"), "w") as fh:
        json.dump({"hidden_size": 768, "num_attention_heads": 12, "num_hidden_layers": 2, "rms_norm_eps": 1e-5}, fh)
    rep = va.check_assets(root)
    assert rep["ok"] and rep["files"]["gpt"]["layers"] == 2 and rep["files"]["dvae"].get("absent")
    assert rep["gpt_config"]["runtime_fields"]["rms_eps"] == 1e-5
    assert all(not rep["files"][n]["missing"] and not rep["files"][n]["unexpected"] for n in W.ASSET_FILES)
    assert va.main([root, "--json", "--no-pins"]) == 0
    for name, fn in (("vocos", lambda: va.pin_vocos(weights["vocos"])), ("vector_quantize_pytorch", lambda: va.pin_gfsq(W.synthetic_dvae())),
                     ("torchaudio", va.pin_mel)):
        if importlib.util.find_spec(name) is None:
            assert fn()["status"] == "absent"
    bad = {(k.replace("gamma", "layer_scale") if k.startswith("backbone.convnext.0.g") else k): v for k, v in weights["vocos"].items()}
    bad["head.out.weight"] = torch.zeros(1026, 256)
    bad["feature_extractor.mel_spec.mel_scale.fb"] = torch.zeros(513, 100)       # present in the real file, not read by the hot path
    W.save_assets(os.path.join(root, "asset"), {"vocos": bad})
    rep = va.check_assets(root)
    v = rep["files"]["vocos"]
    assert not rep["ok"] and not v["ok"] and v["missing"] == ["backbone.convnext.0.gamma"] and v["unexpected"] == ["backbone.convnext.0.layer_scale"]
    assert v["ignored"] == ["feature_extractor.mel_spec.mel_scale.fb"] and "(1026, 256) != expected (1026, 512)" in v["wrong_shape"][0]
    assert va.main([root, "--no-pins"]) == 1
    with open(os.path.join(root, "asset", "gpt", "config.json"), "w") as fh:
        json.dump({"hidden_size": 1024}, fh)
    assert not va.check_assets(root)["gpt_config"]["ok"]


def test_left_pad_starts():
    m = torch.tensor([[0, 0, 1, 1], [1, 1, 1, 1], [0, 1, 1, 1]])
    assert E.left_pad_starts(m).tolist() == [2, 0, 1]
    with pytest.raises(NotImplementedError):
        E.left_pad_starts(torch.tensor([[1, 0, 1, 1]]))
    with pytest.raises(ValueError):
        E.left_pad_starts(torch.tensor([[0, 0, 0, 0]]))


def test_shard_bounds_cover():
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            b = [D.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_exp_draws_shard_is_slice_of_full_batch():
    full = rng.ExpDraws(64, 626, 42).step(0)
    part = rng.ExpDraws(64, 626, 42, row_begin=24, row_end=40).step(0)
    assert torch.equal(part, full[24:40])
    assert torch.equal(rng.ExpDraws(64, 626, 42).step(5), full)  # re-seeded every step (gpt.py:504-507)
    torch.manual_seed(3)
    a = rng.ExpDraws(8, 626, None)
    s0, s1 = a.step(0), a.step(1)
    assert not torch.equal(s0, s1)
    torch.manual_seed(3)
    assert torch.equal(torch.empty(8, 626).exponential_(1), s0)


def test_rope_table_matches_hf():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    cfg = LlamaConfig(hidden_size=768, num_attention_heads=12, max_position_embeddings=4096)
    rot = LlamaRotaryEmbedding(cfg)
    pos = torch.arange(0, 300)[None]
    cos, sin = rot(torch.zeros(1, 300, 768), pos)
    c, s = E.rope_tables(300)
    assert torch.equal(cos[0, :, :32], c) and torch.equal(sin[0, :, :32], s)
    assert torch.equal(cos[0, :, 32:], c)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from chattts_amd import weights as W
        sds = None
        if rank == 0:
            sds = {"gpt": W.synthetic_gpt(n_layers=1), "embed": {"emb_code.0.weight": torch.randn(626, 8)},
                   "decoder": {"coef": torch.rand(1, 100, 1)}, "vocos": {"head.istft.window": torch.hann_window(1024)}}
        meta = {"gpt": D.weights_meta(1)["gpt"], "embed": {"emb_code.0.weight": ((626, 8), torch.float32)},
                "decoder": {"coef": ((1, 100, 1), torch.float32)}, "vocos": {"head.istft.window": ((1024,), torch.float32)}}
        got = D.broadcast_state_dicts(sds, src=0, device=None, meta=meta)
        fp = W.fingerprint(got["gpt"])
        # sharded sampling == rows of the full batch (global row numbering for draws and the >=625 quirk)
        from oracle import cases, sampling_np
        c = cases.SAMPLING_CASES["rows640"]
        logits, hist, temp = cases.sampling_inputs(c)
        lo, hi = D.shard_bounds(160, world, rank)
        r0, r1 = lo * 4, hi * 4
        qd = rng.ExpDraws(640, 626, c["seed"], row_begin=r0, row_end=r1).step(0).numpy()
        idx = sampling_np.sample_step(logits[r0:r1], hist[r0:r1], qd, temperature=temp[r0:r1], top_p=c["top_P"], top_k=c["top_K"],
                                      pow_table=rng.penalty_table(c["rep"]).numpy(), max_input_ids=625, row_offset=r0)
        gathered = [None] * world
        dist.all_gather_object(gathered, (fp, r0, idx.tolist()))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_and_sharded_sampling(golden):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0]  # identical weights on both ranks after the broadcast
    idx = np.array(res[0][2] + res[1][2])
    assert np.array_equal(idx, golden["sampling"]["rows640.idx"])  # union of shards == the reference's full-batch run


def _bench_shard_worker(rank, world, port, q):
    """one rank of bench.py's data-parallel path on CPU: bench.shard_workload -> this shard's prompts / forced lengths /
    global row numbering -> generation (numpy oracle standing in for the HIP engine) -> gather in rank order"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from chattts_amd import weights as W
        from oracle import generate_np, llama_np
        wl = bench.shard_workload(3, world, rank, 4, 9)      # 3 utterances per rank
        llama = llama_np.LlamaWeights({k: v.numpy() for k, v in W.synthetic_gpt(n_layers=2).items()})
        esd = {k: v.numpy() for k, v in W.synthetic_embed().items()}
        srows = (np.asarray(wl["sel"])[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)      # global sampling rows of this shard
        draws = rng.ExpDraws(wl["total_rows"], 626, 42, rows=torch.from_numpy(srows))
        res = generate_np.generate(llama, esd, generate_np.fold_heads(esd), generate_np.embed_prompt(esd, wl["ids"], wl["tmask"]),
                                   wl["ids"], wl["mask"], temperature=np.array([0.3] * 4, np.float32), draw_q=lambda i: draws.step(i).numpy(),
                                   pow_table=rng.penalty_table(1.05).numpy(), max_new_token=int(wl["stop_all"].max()) + 1,
                                   stop_at=wl["stop"], row_offset=srows)
        gathered = [None] * world
        dist.all_gather_object(gathered, (list(wl["sel"]), [r.tolist() for r in res.ids]))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_rank_bench_sharding_path_equals_the_unsharded_batch():
    """bench.py --gpus 2 on CPU/gloo: the two ranks' shards (dealt by prompt length, dist.deal_shards), put back in the caller's order,
    are the single-process run of the global batch row for row (draws and the rows>=625 quirk keyed on the global row index)"""
    import torch.multiprocessing as mp
    import bench
    from chattts_amd import weights as W
    from oracle import generate_np, llama_np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][0] + res[1][0]) == list(range(6)) and len(res[0][0]) == len(res[1][0]) == 3
    assert res[0][0] != [0, 1, 2]                            # (dealt by prompt length: the shards are not contiguous blocks)
    sharded = [None] * 6
    for sel, rows in res:
        for b, r in zip(sel, rows):
            sharded[b] = np.array(r, dtype=np.int64).reshape(-1, 4)
    wl = bench.shard_workload(6, 1, 0, 4, 9)                 # the same 6 utterances as ONE batch
    llama = llama_np.LlamaWeights({k: v.numpy() for k, v in W.synthetic_gpt(n_layers=2).items()})
    esd = {k: v.numpy() for k, v in W.synthetic_embed().items()}
    draws = rng.ExpDraws(24, 626, 42)
    full = generate_np.generate(llama, esd, generate_np.fold_heads(esd), generate_np.embed_prompt(esd, wl["ids"], wl["tmask"]), wl["ids"],
                                wl["mask"], temperature=np.array([0.3] * 4, np.float32), draw_q=lambda i: draws.step(i).numpy(),
                                pow_table=rng.penalty_table(1.05).numpy(), max_new_token=int(wl["stop_all"].max()) + 1, stop_at=wl["stop"])
    assert [len(r) for r in sharded] == wl["stop"].tolist()
    for a, b in zip(sharded, full.ids):
        assert np.array_equal(a, b)


class _OracleChat:
    """the two seams `dist.infer_sharded` drives (Chat.infer_code / Chat.decode_to_wavs), served by the numpy oracle on a 2-layer slice of
    the architecture and by a stand-in decoder whose output depends on what the real one depends on: the rows AND the padded width"""

    def __init__(self):
        from chattts_amd import weights as W
        from oracle import generate_np, llama_np
        self.llama = llama_np.LlamaWeights({k: v.numpy() for k, v in W.synthetic_gpt(n_layers=2).items()})
        self.esd = {k: v.numpy() for k, v in W.synthetic_embed().items()}
        self.heads = generate_np.fold_heads(self.esd)
        self.calls = []

    def infer_code(self, ids, mask, tmask, params, stream=False, return_hidden=True, row_ids=None, total_rows=None, stop_at=None):
        from types import SimpleNamespace
        from oracle import generate_np
        B = ids.shape[0]
        gl = np.arange(B) if row_ids is None else np.asarray(row_ids)
        self.calls.append(gl.tolist())
        srows = (gl[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
        draws = rng.ExpDraws(total_rows if total_rows is not None else B * 4, 626, 42, rows=torch.from_numpy(srows))
        res = generate_np.generate(self.llama, self.esd, self.heads, generate_np.embed_prompt(self.esd, ids.numpy(), tmask.numpy()), ids.numpy(),
                                   mask.numpy(), temperature=np.array([0.3] * 4, np.float32), draw_q=lambda i: draws.step(i).numpy(),
                                   pow_table=rng.penalty_table(1.05).numpy(), max_new_token=12, stop_at=None if stop_at is None else stop_at.numpy(),
                                   row_offset=srows)
        yield SimpleNamespace(ids=[torch.from_numpy(r) for r in res.ids], hiddens=[torch.from_numpy(h) for h in res.hiddens])

    def decode_to_wavs(self, rows, use_decoder=True, pad_to=None):
        T = max(int(r.shape[0]) for r in rows) if pad_to is None else int(pad_to)
        wav = np.full((len(rows), 4 * T), 0.25, np.float32)       # the zero-padded tail of a shorter row is NOT silent in the reference
        for i, r in enumerate(rows):
            wav[i, : 4 * r.shape[0]] = np.repeat(r.numpy().astype(np.float32).sum(1), 4)
        return wav


def _sharded_inputs():
    from chattts_amd import synth
    ids, mask, tmask = synth.make_prompts(7, 4, 9, seed=3)          # 7 utterances over 2 ranks: uneven shards
    stop = synth.make_stop_lengths(7, 3, 10, seed=3)
    return torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask), torch.from_numpy(stop)


def _infer_sharded_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chat = _OracleChat()
        ids, mask, tmask, stop = _sharded_inputs()
        got = D.infer_sharded(chat, ids, mask, tmask, None, stop_at=stop, return_ids=True)
        own = D.infer_sharded(chat, ids, mask, tmask, None, stop_at=stop, gather=False)
        if rank == 0:
            q.put((got[0], [r.tolist() for r in got[1]], chat.calls, own[0]))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_infer_sharded_over_two_ranks_equals_the_single_process_call():
    """round 6: `dist.infer_sharded` -- deal by prompt length, generate with global row ids, ONE all-reduce(max) of the
    longest utterance, decode padded to it, gather to rank 0 in the caller's order -- over 2 gloo ranks == the same call in one process
    (== the plain unsharded sequence of seam calls), waveforms AND token ids; gather=False hands every rank its own utterances."""
    import torch.multiprocessing as mp
    shards = D.deal_shards([5, 9, 9, 4, 7, 7, 6], 2)
    assert shards == [[1, 5, 6], [0, 2, 3, 4]] or sorted(shards[0] + shards[1]) == list(range(7))
    assert D.deal_shards([3, 1, 2], 1) == [[0, 1, 2]] and D.deal_shards([1, 2, 3, 4], 2, "blocks") == [[0, 1], [2, 3]]
    assert D.deal_shards([1, 2, 3, 4], 2, "sorted_blocks") == [[2, 3], [0, 1]]
    chat = _OracleChat()
    ids, mask, tmask, stop = _sharded_inputs()
    full, full_ids = D.infer_sharded(chat, ids, mask, tmask, None, stop_at=stop, return_ids=True)        # no process group: world 1
    assert chat.calls == [list(range(7))]
    direct = list(chat.infer_code(ids, mask, tmask, None, stop_at=stop))[-1]
    assert np.array_equal(full, chat.decode_to_wavs(direct.hiddens)) and [len(r) for r in full_ids] == stop.tolist()
    # the code-book path (use_decoder=False, core.py:518,535): the token-id rows travel through the same recipe
    assert np.array_equal(D.infer_sharded(chat, ids, mask, tmask, None, stop_at=stop, use_decoder=False), chat.decode_to_wavs(direct.ids, False))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_infer_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    wav2, ids2, calls, own0 = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lens = mask.sum(1).tolist()
    assert calls[0] == D.deal_shards(lens, 2)[0] == own0 and calls[0] != list(range(len(calls[0])))
    assert wav2.shape == full.shape and np.array_equal(wav2, full)
    assert all(np.array_equal(np.asarray(a), b) for a, b in zip(ids2, full_ids))


def _verdict_worker(rank, world, port, q):
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_c3_w2.npz"))
        off = np.concatenate([[0], np.cumsum(gold["lens"].astype(np.int64))])
        wl = bench.shard_workload(64, world, rank, 128, 512)
        gold_rows = [gold["ids"][off[b]: off[b + 1]].astype(np.int64) for b in wl["sel"]]
        same = bench.reference_verdict([r.copy() for r in gold_rows], gold_rows, wl["sel"], [], dist, world)
        rows = [r.copy() for r in gold_rows]
        if rank == 0:
            rows[3][5, 2] += 1                              # one token of this rank's 4th utterance
        flagged = bench.reference_verdict(rows, gold_rows, wl["sel"], [3, 7] if rank == 0 else [1], dist, world)
        unflagged = bench.reference_verdict(rows, gold_rows, wl["sel"], [7] if rank == 0 else [], dist, world)
        nocert = bench.reference_verdict(rows, gold_rows, wl["sel"], None, dist, world)
        q.put((rank, same, flagged, unflagged, nocert, int(wl["sel"][3])))
    finally:
        dist.destroy_process_group()


def test_bench_reference_verdict_over_two_ranks():
    """`bench.reference_verdict` (what `ids_check.ids_match_reference` of an N-rank line is made of): every rank compares ITS shard's token
    rows with the reference's rows of the same utterances of the N = 2 global batch (tests/golden/bench_c3_w2.npz, the reference's own run),
    one all_gather_object makes the verdict the same on every rank -- the AND, the global indices of the differing utterances, and whether
    the certificate had flagged every one of them."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verdict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = got[0][5]                                            # global index of rank 0's 4th utterance
    for rank, same, flagged, unflagged, nocert, _ in got:
        assert same == (True, {})
        assert flagged == (False, {"differing_utterances": [b], "utterances_compared": 128, "differing_all_flagged_by_certificate": True})
        assert unflagged[0] is False and unflagged[1]["differing_all_flagged_by_certificate"] is False
        assert nocert[0] is False and nocert[1]["differing_all_flagged_by_certificate"] is None

def _run_bench(*argv, env_drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env["HIP_VISIBLE_DEVICES"] = ""        # this test is about the launcher: no GPU, whatever the box has
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the command: the entry spawns two ranks itself; with
    --plumbing-only they rendezvous on gloo here, shard the 128-utterance global batch in contiguous blocks and end up with the same
    broadcast weights -- `ranks.world == 2` on the one line rank 0 prints."""
    import json
    r = _run_bench("--gpus", "2", "--plumbing-only")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["plumbing_only"] is True and j["value"] is None            # cannot be mistaken for a measurement
    assert j["n_gpus"] == 2 and j["ranks"]["world"] == 2 and j["global_batch"] == 128
    shards = [s["utterances"] for s in j["ranks"]["shards"]]
    assert [len(x) for x in shards] == [64, 64] and sorted(shards[0] + shards[1]) == list(range(128))
    assert all(x == sorted(x) for x in shards) and shards[0] != list(range(64))           # dealt by prompt length, not contiguous blocks
    pt = [s["prompt_tokens"] for s in j["ranks"]["shards"]]
    assert abs(pt[0] - pt[1]) <= 48                                                       # ... which balances the ranks' prompt tokens
    assert j["ranks"]["weights_equal_on_all_ranks"] and j["ranks"]["rows_cover_global_batch"]


def test_bench_gpus_2_without_gpus_fails_loudly_about_the_gpu():
    """... and the real thing on a box without GPUs: the spawned ranks say which GPU they miss, the launcher exits non-zero and NO
    result line is printed -- never a 1-rank number labelled n_gpus = 2."""
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), r.stdout
    # (the launcher tears the other rank down as soon as the first one exits: at least one of them got to say which GPU it misses)
    assert ("needs GPU 1 of 2" in r.stderr or "needs GPU 0 of 2" in r.stderr) and "no CPU fallback" in r.stderr
    assert "NO result line is printed for --gpus 2" in r.stderr


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to report a 1-rank run as n_gpus=8" in r.stderr


def test_pmc_summary_maps_the_profiled_kernel_names():
    """tools/pmc_summary.py maps rocprofv3 kernel names to bench.py's tags by substring; the committed kernel-stat CSV of the
    round must still resolve to every tag bench.py's roofline leg can ask for (catches template-argument drift)."""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(ROOT, "profiles", "r3b_kernel_stats_bf16.csv")) as f:
        names = [r["Name"] for r in csv.DictReader(f)]
    tags = {mod.short(n) for n in names} - {None}
    assert {"attention", "qkv_gemm", "o_proj_gemm", "gate_up_gemm", "down_gemm", "heads_gemm", "sample"} <= tags
    with open(os.path.join(ROOT, "profiles", "r3b_kernel_stats_f32.csv")) as f:      # the parity mode's own roofline entry
        assert "attention_f32" in {mod.short(r["Name"]) for r in csv.DictReader(f)}
    import json
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        traffic = json.load(f)
    assert traffic["attention"]["hbm_bytes_per_launch"] > 1e7 and traffic["attention"]["dispatches"] > 1000
    assert traffic["attention_f32"]["hbm_bytes_per_launch"] > 2e7 and traffic["heads_gemm"]["dispatches"] > 100
    with open(os.path.join(ROOT, "profiles", "r3ak_kernel_stats.csv")) as f:         # the final tree's one-pass profile: decoder kernels too
        tags = {mod.short(r["Name"]) for r in csv.DictReader(f)} - {None}
    assert {"attention", "qkv_gemm", "o_proj_gemm", "gate_up_gemm", "down_gemm", "heads_gemm", "sample", "codec_pwconv1_h1p",
            "codec_pwconv2_h1p", "dwconv_ln_run"} <= tags
    assert {"codec_pwconv1_h1p", "codec_pwconv2_h1p", "dwconv_ln_run"} <= set(traffic)


def test_pipelined_queue_yields_one_item_per_batch_in_order():
    """`Chat.infer_ids_pipelined`: exactly one result per input batch, in order -- also when a batch in the middle of the queue produces
    no output (seeded generation that ends at step 0 yields nothing, gpt.py:570): its empty placeholder must not be dropped, or every
    later result would be attributed to the wrong batch.  Host logic only: generation and the acoustic decoder are stubbed."""
    import types

    import numpy as np

    from chattts_amd.core import Chat

    class Pend:
        def __init__(self, v):
            self.v = v

        def result(self):
            return np.full((1, 2), self.v, np.float32)

    chat = Chat.__new__(Chat)
    chat.codec = types.SimpleNamespace(decode_to_wavs_async=lambda hid: Pend(hid))
    outs = {"a": 1.0, "b": None, "c": 3.0, "d": None, "e": None, "f": 6.0}

    def infer_code(ids, mask, tmask, params, stream=False, **kw):
        if outs[ids] is not None:
            yield types.SimpleNamespace(hiddens=outs[ids])

    chat.infer_code = infer_code
    got = list(chat.infer_ids_pipelined([(k, None, None) for k in outs]))
    assert len(got) == len(outs)
    for g, k in zip(got, outs):
        if outs[k] is None:
            assert g.shape == (0,)
        else:
            assert g.shape == (1, 2) and float(g[0, 0]) == outs[k]
