"""Per-kernel parity: every HIP kernel, called through the C ABI, against the numpy oracle
(oracle/*.py) or the reference-generated goldens.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chattts_amd import _lib, rng  # noqa: E402
from oracle import cases, codec_np, llama_np, sampling_np  # noqa: E402

f32 = np.float32


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return gpu_util


# ------------------------------------------------------------------------------------------------
# skinny GEMM (decode / prefill projections)
# ------------------------------------------------------------------------------------------------
SK_SHAPES = [  # (N, K, epi, rms)
    (2304, 768, 0, True),    # RMSNorm + QKV
    (768, 768, 1, False),    # o_proj + residual
    (3072, 768, 2, True),    # RMSNorm + gate/up + SiLU*up
    (768, 3072, 1, False),   # down_proj + residual
    (2504, 768, 0, False),   # heads (N not a multiple of 16)
]


@pytest.mark.parametrize("wt", ["f32", "bf16"])
@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 150])
@pytest.mark.parametrize("shape", SK_SHAPES)
def test_gemm_skinny(G, wt, M, shape):
    N, K, epi, rms = shape
    rs = np.random.RandomState(M * 131 + N + K)
    A = rs.standard_normal((M, K)).astype(f32) * (2.0 if rms else 1.0)
    nrows = 2 * N if epi == 2 else N
    W = (rs.standard_normal((nrows, K)) * 0.03).astype(f32)
    nw = (1.0 + 0.1 * rs.standard_normal(K)).astype(f32) if rms else None
    res = rs.standard_normal((M, N)).astype(f32) if epi == 1 else None
    got = G.gemm(A, W, wt=wt, epi=epi, norm_w=nw, res=res, n_out=N)
    An = llama_np.rmsnorm(A, nw, f32(1e-6)) if rms else A
    Wr = W
    if wt == "bf16":
        An, Wr = G.bf16_round(An), G.bf16_round(W)
    acc = An.astype(np.float64) @ Wr.astype(np.float64).T
    if epi == 2:
        g, u = acc[:, :N], acc[:, N:]
        ref = g / (1.0 + np.exp(-g)) * u
    elif epi == 1:
        ref = res + acc
    else:
        ref = acc
    assert np.isfinite(got).all()
    tol = 3e-6 if wt == "f32" else 2e-4   # bf16: inputs are rounded identically; only f32 accumulation order differs
    assert G.relerr(got, ref) < tol, (wt, M, shape, G.relerr(got, ref))


@pytest.mark.parametrize("M", [1, 9, 16, 31, 64, 200, 300, 1111, 1500, 11008])   # >= 256 rows: the LDS-tiled prefill kernel (64- and 128-row tiles)
def test_gemm_fast_bf16(G, M):
    """perf-mode projections: bf16 activations, row scale from partial sums of squares, three epilogues"""
    lib = _lib.lib()
    rs = np.random.RandomState(M)
    x32 = (rs.standard_normal((M, 768)) * 2).astype(f32)
    x_d = G.dev(x32)
    xb = torch.empty((M, 768), dtype=torch.bfloat16, device=G.DEV)
    ssq = torch.empty((M, 48), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_rows_prep(x_d.data_ptr(), xb.data_ptr(), ssq.data_ptr(), M, None), "rows_prep")
    xbr = G.bf16_round(x32)
    assert np.array_equal(xb.float().cpu().numpy(), xbr)
    assert np.abs(ssq.cpu().numpy().sum(1) - (x32.astype(np.float64) ** 2).sum(1)).max() < 1e-2
    rstd = 1.0 / np.sqrt((x32.astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
    # epi 0: (x_bf16 @ W^T) * rstd
    W = G.bf16_round((rs.standard_normal((2304, 768)) * 0.03).astype(f32))
    W_d = G.dev(W, torch.bfloat16)
    C = torch.full((M, 2304), float("nan"), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_gemm_fast(xb.data_ptr(), 768, W_d.data_ptr(), M, 2304, 768, ssq.data_ptr(), 1e-6, 0, C.data_ptr(), 2304,
                                    None, 0, None, None), "fast0")
    ref = (xbr.astype(np.float64) @ W.astype(np.float64).T) * rstd
    assert G.relerr(C.cpu().numpy(), ref) < 2e-4
    # epi 2: silu(g) * u -> bf16
    Wgu = G.bf16_round((rs.standard_normal((2 * 3072, 768)) * 0.03).astype(f32))
    Wgu_d = G.dev(Wgu, torch.bfloat16)
    act = torch.zeros((M, 3072), dtype=torch.bfloat16, device=G.DEV)
    _lib.check(lib.ctts_k_gemm_fast(xb.data_ptr(), 768, Wgu_d.data_ptr(), M, 3072, 768, ssq.data_ptr(), 1e-6, 2, None, 0,
                                    act.data_ptr(), 3072, None, None), "fast2")
    gu = (xbr.astype(np.float64) @ Wgu.astype(np.float64).T) * rstd
    g_, u_ = gu[:, :3072], gu[:, 3072:]
    refa = g_ / (1 + np.exp(-g_)) * u_
    assert G.relerr(act.float().cpu().numpy(), refa) < 1e-2  # bf16 output rounding
    # epi 1 with K = 3072 (down_proj): residual update + bf16 copy + new partial sums
    Wd = G.bf16_round((rs.standard_normal((768, 3072)) * 0.02).astype(f32))
    Wd_d = G.dev(Wd, torch.bfloat16)
    res = x_d.clone()
    xb2 = torch.empty_like(xb)
    ssq2 = torch.empty_like(ssq)
    _lib.check(lib.ctts_k_gemm_fast(act.data_ptr(), 3072, Wd_d.data_ptr(), M, 768, 3072, None, 0.0, 1, res.data_ptr(), 768,
                                    xb2.data_ptr(), 768, ssq2.data_ptr(), None), "fast1")
    actf = act.float().cpu().numpy().astype(np.float64)
    refx = x32 + actf @ Wd.astype(np.float64).T
    got = res.cpu().numpy()
    assert G.relerr(got, refx) < 2e-5
    assert np.array_equal(xb2.float().cpu().numpy(), G.bf16_round(got))
    assert np.abs(ssq2.cpu().numpy().sum(1) - (got.astype(np.float64) ** 2).sum(1)).max() < 1e-2
    # epi 1 with K = 768 (o_proj)
    Wo = G.bf16_round((rs.standard_normal((768, 768)) * 0.03).astype(f32))
    res = x_d.clone()
    _lib.check(lib.ctts_k_gemm_fast(xb.data_ptr(), 768, G.dev(Wo, torch.bfloat16).data_ptr(), M, 768, 768, None, 0.0, 1, res.data_ptr(),
                                    768, xb2.data_ptr(), 768, ssq2.data_ptr(), None), "fast1o")
    assert G.relerr(res.cpu().numpy(), x32 + xbr.astype(np.float64) @ Wo.astype(np.float64).T) < 2e-5


@pytest.mark.parametrize("force_mb", [0, 1, 2, 4])
@pytest.mark.parametrize("M,n_act", [(1, None), (9, None), (16, 16), (40, 33), (64, None), (64, 45), (100, 70), (130, None)])
def test_gemm_dec_packed(G, M, n_act, force_mb):
    """decode projections on fragment-packed operands (csrc/decode.hip): SiLU and residual epilogues, both K, partial row
    tiles, the live-row count read from the device (rows >= n_active are neither computed nor written)"""
    from chattts_amd.engine import pack_frag, unpack_frag
    lib = _lib.lib()
    rs = np.random.RandomState(M * 7 + (n_act or 0))
    Mp = (M + 15) // 16 * 16
    live = M if n_act is None else n_act
    na_d = None if n_act is None else G.dev(np.array([n_act], np.int32))
    x32 = (rs.standard_normal((M, 768)) * 2).astype(f32)
    xbr = G.bf16_round(x32)
    xpad = np.zeros((Mp, 768), f32)
    xpad[:M] = xbr
    xp = pack_frag(torch.from_numpy(xpad).to(torch.bfloat16)).to(G.DEV)
    ssq_np = (x32.astype(np.float64) ** 2).reshape(M, 48, 16).sum(2).astype(f32)
    ssq = G.dev(ssq_np)
    rstd = 1.0 / np.sqrt((x32.astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
    # epi 2: silu(g) * u -> packed bf16 [Mp][3072]
    Wgu = G.bf16_round((rs.standard_normal((2 * 3072, 768)) * 0.03).astype(f32))
    Wgu_p = pack_frag(torch.from_numpy(Wgu).to(torch.bfloat16)).to(G.DEV)
    actp = torch.full((Mp * 3072,), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    _lib.check(lib.ctts_k_gemm_dec(xp.data_ptr(), Wgu_p.data_ptr(), M, 3072, 768, _lib.ptr(na_d), ssq.data_ptr(), 1e-6, 2, None, 0,
                                   actp.data_ptr(), 96, None, force_mb, None), "dec silu")
    torch.cuda.synchronize()
    gu = (xbr.astype(np.float64) @ Wgu.astype(np.float64).T) * rstd
    g_, u_ = gu[:, :3072], gu[:, 3072:]
    refa = g_ / (1 + np.exp(-g_)) * u_
    act = unpack_frag(actp.float().cpu(), Mp, 3072).numpy()
    assert G.relerr(act[:live], refa[:live]) < 1e-2  # bf16 output rounding
    assert np.isnan(act[live:M]).all()                # rows beyond the live count are not written
    # epi 1, K = 3072 (down_proj): residual update in place + packed bf16 copy + partial sums of squares
    actc = np.where(np.isnan(act), 0, act).astype(f32)
    actp2 = pack_frag(torch.from_numpy(actc).to(torch.bfloat16)).to(G.DEV)
    Wd = G.bf16_round((rs.standard_normal((768, 3072)) * 0.02).astype(f32))
    Wd_p = pack_frag(torch.from_numpy(Wd).to(torch.bfloat16)).to(G.DEV)
    res = G.dev(x32).clone()
    xp2 = torch.full((Mp * 768,), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    ssq2 = torch.full((M, 48), float("nan"), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_gemm_dec(actp2.data_ptr(), Wd_p.data_ptr(), M, 768, 3072, _lib.ptr(na_d), None, 0.0, 1, res.data_ptr(), 768,
                                   xp2.data_ptr(), 24, ssq2.data_ptr(), force_mb, None), "dec down")
    torch.cuda.synchronize()
    refx = x32 + actc[:M].astype(np.float64) @ Wd.astype(np.float64).T
    got = res.cpu().numpy()
    assert G.relerr(got[:live], refx[:live]) < 2e-5
    assert np.array_equal(got[live:], x32[live:])
    xb2 = unpack_frag(xp2.float().cpu(), Mp, 768).numpy()
    assert np.array_equal(xb2[:live], G.bf16_round(got[:live])) and np.isnan(xb2[live:M]).all()
    assert np.abs(ssq2.cpu().numpy()[:live].sum(1) - (got[:live].astype(np.float64) ** 2).sum(1)).max() < 1e-2
    # epi 1, K = 768 (o_proj)
    Wo = G.bf16_round((rs.standard_normal((768, 768)) * 0.03).astype(f32))
    Wo_p = pack_frag(torch.from_numpy(Wo).to(torch.bfloat16)).to(G.DEV)
    res = G.dev(x32).clone()
    _lib.check(lib.ctts_k_gemm_dec(xp.data_ptr(), Wo_p.data_ptr(), M, 768, 768, _lib.ptr(na_d), None, 0.0, 1, res.data_ptr(), 768,
                                   xp2.data_ptr(), 24, ssq2.data_ptr(), force_mb, None), "dec o")
    torch.cuda.synchronize()
    refo = x32 + xbr.astype(np.float64) @ Wo.astype(np.float64).T
    assert G.relerr(res.cpu().numpy()[:live], refo[:live]) < 2e-5


@pytest.mark.parametrize("force_mb", [0, 1, 2, 4])
@pytest.mark.parametrize("M,n_act", [(1, None), (9, None), (16, 16), (40, 33), (64, None), (64, 45), (100, 70)])
def test_gemm_dec32_bit_identical(G, M, n_act, force_mb):
    """parity-mode decode projections on fragment-packed f32 operands (csrc/decode32.hip) against gemm_skinny_k<float>, the
    kernel the reference goldens of the f32 mode were established with: EXACT equality of every output (same operation order
    per element), for the four projections of a layer, partial row tiles and a device-side live-row count"""
    from chattts_amd.engine import pack_frag32, unpack_frag32
    lib = _lib.lib()
    rs = np.random.RandomState(M * 11 + (n_act or 0) + force_mb)
    Mp = (M + 15) // 16 * 16
    live = M if n_act is None else n_act
    na_d = None if n_act is None else G.dev(np.array([n_act], np.int32))

    def packed(a, rows, cols):     # [rows_real, cols] -> packed [rows, cols] (pad rows = NaN: they must never reach a live output)
        pad = np.full((rows, cols), np.nan, f32)
        pad[: a.shape[0]] = a
        return pack_frag32(torch.from_numpy(pad)).to(G.DEV)

    for N, K, epi, rms in SK_SHAPES:      # the last one is the heads GEMM: N = 2504 is padded to 2512 zero rows, n_cols = 2504
        A = (rs.standard_normal((M, K)) * (2.0 if rms else 1.0)).astype(f32)
        nrows = 2 * N if epi == 2 else N
        W = (rs.standard_normal((nrows, K)) * 0.03).astype(f32)
        Np = (N + 15) // 16 * 16
        nw = (1.0 + 0.1 * rs.standard_normal(K)).astype(f32) if rms else None
        res = rs.standard_normal((M, N)).astype(f32) if epi == 1 else None
        want = G.gemm(A[:live], W, wt="f32", epi=epi, norm_w=nw, res=None if res is None else res[:live], n_out=N)   # gemm_skinny_k<float>
        Wpad = W if Np == N else np.concatenate([W, np.zeros((Np - N, K), f32)], 0)
        A_d, Ap, Wp = G.dev(A), packed(A, Mp, K), pack_frag32(torch.from_numpy(Wpad)).to(G.DEV)
        nw_d = None if nw is None else G.dev(nw)
        Cc = torch.full((M, N), float("nan"), dtype=torch.float32, device=G.DEV)
        Cp = torch.full((Mp * N,), float("nan"), dtype=torch.float32, device=G.DEV)
        res_d = None if res is None else G.dev(res)
        _lib.check(lib.ctts_k_gemm_dec32(Ap.data_ptr(), Wp.data_ptr(), M, Np, K, _lib.ptr(na_d), A_d.data_ptr() if rms else None, K,
                                         _lib.ptr(nw_d), 1e-6, epi, Cc.data_ptr(), N, _lib.ptr(res_d), N, Cp.data_ptr() if epi else None,
                                         N // 16, force_mb, N, None), "dec32")
        torch.cuda.synchronize()
        # the variants are bit-identical, so a dispatch regression would be invisible below: pin the intended choices
        variant = lib.ctts_k_dec32_last_variant().decode()
        assert variant == ("generic" if force_mb in (2, 4) else "rms16" if rms else "m16"), (N, K, epi, force_mb, variant)
        got_c = Cc.cpu().numpy()
        got_p = unpack_frag32(Cp.cpu(), Mp, N).numpy() if epi != 0 else None
        if epi != 2:
            assert np.array_equal(got_c[:live], want), (N, K, epi)
            assert np.isnan(got_c[live:]).all()                       # rows beyond the live count are not written
        if epi != 0:
            assert np.array_equal(got_p[:live], want), (N, K, epi)
            assert np.isnan(got_p[live:M]).all()


@pytest.mark.parametrize("M", [3072, 200, 37])
def test_gemm_pre_x3_split_bf16_prefill(G, M):
    """round 6: the LDS-tiled split-fp16 GEMM of the f32x3 mode's prompt pass (csrc/prefill32x.hip) for the four projection shapes of a
    layer -- f32 row-major operands split by the tile loader (hi = fp16(x), lo' = fp16((x - hi) 2^11)), RMSNorm gain applied before the
    split, 1 / rms on the accumulator -- (a) against the float64 value of exactly what it is defined to compute ((hi+lo)(hi+lo) - lo*lo
    over the split operands) within f32 accumulation noise, (b) within 3e-6 of the output's scale of the float64 product of the UNSPLIT
    operands (the bf16 split of round 5 needed 2e-5); 128- and 64-row tiles, a partial last tile."""
    lib = _lib.lib()
    rs = np.random.RandomState(M)

    def split64(a):       # the split-fp16 operand format (engine.split_f16 / common.hpp x3_split): value = hi + lo' / 2048
        from chattts_amd.engine import X3_LO_SCALE, split_f16
        hi, lo = split_f16(torch.from_numpy(a.astype(f32)))
        return hi.double().numpy(), lo.double().numpy() / X3_LO_SCALE

    for N, K, epi, rms in [(2304, 768, 0, True), (768, 768, 1, False), (3072, 768, 2, True), (768, 3072, 1, False)]:
        A = (rs.standard_normal((M, K)) * (2.0 if rms else 1.0)).astype(f32)
        nrows = 2 * N if epi == 2 else N
        Wm = (rs.standard_normal((nrows, K)) * 0.03).astype(f32)
        g = (1.0 + 0.1 * rs.standard_normal(K)).astype(f32) if rms else None
        rstd = (1.0 / np.sqrt((A.astype(np.float64) ** 2).mean(1) + 1e-6)).astype(f32) if rms else None
        res = rs.standard_normal((M, N)).astype(f32) if epi == 1 else None
        C_d = torch.full((M, N), float("nan"), dtype=torch.float32, device=G.DEV)
        keep = [G.dev(A), G.dev(Wm)] + [None if t is None else G.dev(t) for t in (g, rstd, res)]
        _lib.check(lib.ctts_k_gemm_pre_x3(keep[0].data_ptr(), K, keep[1].data_ptr(), C_d.data_ptr(), N, M, N, K, epi, _lib.ptr(keep[2]),
                                          _lib.ptr(keep[3]), _lib.ptr(keep[4]), N, None), "gemm_pre_x3")
        torch.cuda.synchronize()
        got = C_d.cpu().numpy().astype(np.float64)
        Ag = (A * g[None, :]).astype(f32) if rms else A          # the loader's f32 multiply, then the split
        ah, al = split64(Ag)
        wh, wl = split64(Wm)
        acc_def = (ah + al) @ (wh + wl).T - al @ wl.T
        acc_full = Ag.astype(np.float64) @ Wm.astype(np.float64).T
        outs = []
        for acc in (acc_def, acc_full):
            if rms:
                acc = acc * rstd.astype(np.float64)[:, None]
            if epi == 1:
                acc = res.astype(np.float64) + acc
            elif epi == 2:
                gate, up = acc[:, :N], acc[:, N:]
                acc = gate / (1.0 + np.exp(-gate)) * up
            outs.append(acc)
        scale = np.abs(outs[1]).max()
        assert np.isfinite(got).all()
        assert np.abs(got - outs[0]).max() < 3e-6 * scale, (M, N, K, epi, np.abs(got - outs[0]).max() / scale)
        assert np.abs(got - outs[1]).max() < 3e-6 * scale, (M, N, K, epi, np.abs(got - outs[1]).max() / scale)   # 22-bit operands: f32-class


@pytest.mark.parametrize("force_mb", [0, 1, 2, 4, 9, 10, 12, 17])  # rows per workgroup (x 16); + 8: eight waves split K instead of four; 17: sixteen (down_proj only, else eight)
@pytest.mark.parametrize("M,n_act", [(64, None), (64, 37), (16, None), (5, None), (33, 20), (48, 48)])
def test_gemm_dec32x_split_bf16(G, M, n_act, force_mb):
    """parity-mode decode projections on SPLIT-fp16 operands (csrc/decode32x.hip: hi | lo' fp16 planes, value = hi + lo' / 2048, three
    fp16 MFMAs per product, f32 accumulation) for the o_proj / gate-up / down shapes of a layer: (a) against the float64 value of EXACTLY
    what the kernel is defined to compute -- (hi+lo)(hi+lo) - lo*lo over the split operands, 1 / rms from the f32 rows, residual / SiLU
    epilogue -- within f32 accumulation noise; (b) within 5e-6 (relative to the output's scale) of the float64 product of the UNSPLIT
    operands: 22-bit operands (the bf16 split of round 5 needed 2e-5); the output planes re-assemble to the row-major result; partial row tiles,
    a device-side live-row count, rows beyond it untouched."""
    from chattts_amd.engine import pack_frag, pack_frag_x3, unpack_frag, unpack_frag32
    lib = _lib.lib()
    rs = np.random.RandomState(M * 13 + (n_act or 0) + force_mb)
    Mp = (M + 15) // 16 * 16
    live = M if n_act is None else n_act
    na_d = None if n_act is None else G.dev(np.array([n_act], np.int32))

    from chattts_amd.engine import X3_LO_SCALE, split_f16

    def split(a):         # the split-fp16 operand format: value = hi + lo' / 2048 (NaN pad rows stay NaN)
        return split_f16(torch.from_numpy(a))

    for N, K, epi, rms in [(768, 768, 1, False), (3072, 768, 2, True), (768, 3072, 1, False)]:
        A = (rs.standard_normal((M, K)) * (2.0 if rms else 1.0)).astype(f32)
        nrows = 2 * N if epi == 2 else N
        Wm = (rs.standard_normal((nrows, K)) * 0.03).astype(f32)
        res = rs.standard_normal((M, N)).astype(f32) if epi == 1 else None
        Apad = np.full((Mp, K), np.nan, f32)          # pad rows = NaN: they must never reach a live output
        Apad[:M] = A
        ah, al = split(Apad)
        planes_a = torch.stack([pack_frag(ah), pack_frag(al)], 0).contiguous().to(G.DEV)
        planes_w = pack_frag_x3(torch.from_numpy(Wm)).to(G.DEV)
        Cc = torch.full((M, N), float("nan"), dtype=torch.float32, device=G.DEV)
        Cp = torch.full((2, Mp * N), float("nan"), dtype=torch.float32, device=G.DEV).to(torch.bfloat16)
        Cp32 = torch.full((Mp * N,), float("nan"), dtype=torch.float32, device=G.DEV) if epi == 1 else None
        A_d = G.dev(A)
        res_d = None if res is None else G.dev(res)
        # RMSNorm launches: 1 / rms from the rows themselves (even seeds) or from 48 partial sums of squares per row (odd: the decode step's way)
        via_ssq = rms and (M + force_mb) % 2 == 1
        ssq_in = G.dev((A.reshape(M, 48, 16).astype(np.float64) ** 2).sum(-1).astype(f32)) if via_ssq else None
        ssq_out = torch.full((M, 48), float("nan"), dtype=torch.float32, device=G.DEV) if epi == 1 else None
        _lib.check(lib.ctts_k_gemm_dec32x(planes_a.data_ptr(), Mp * K, planes_w.data_ptr(), nrows * K, M, N, K, _lib.ptr(na_d),
                                          A_d.data_ptr() if (rms and not via_ssq) else None, K, 1e-6, epi, Cc.data_ptr() if epi == 1 else None, N,
                                          _lib.ptr(res_d), N, Cp.data_ptr(), Mp * N, N // 32, _lib.ptr(Cp32), force_mb, _lib.ptr(ssq_in),
                                          _lib.ptr(ssq_out), None), "dec32x")
        torch.cuda.synchronize()
        # float64 models
        a_h, a_l = (x.float().numpy().astype(np.float64)[:live] for x in split(A))
        w_h, w_l = (x.float().numpy().astype(np.float64) for x in split(Wm))
        a_l, w_l = a_l / X3_LO_SCALE, w_l / X3_LO_SCALE
        exact = (a_h + a_l) @ (w_h + w_l).T - a_l @ w_l.T          # what three MFMAs sum
        full = A[:live].astype(np.float64) @ Wm.astype(np.float64).T
        if rms:
            rstd = 1.0 / np.sqrt((A[:live].astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
            exact, full = exact * rstd, full * rstd
        if epi == 2:
            sil = lambda v: v / (1.0 + np.exp(-v))
            exact, full = sil(exact[:, :N]) * exact[:, N:], sil(full[:, :N]) * full[:, N:]
        else:
            exact, full = exact + res[:live], full + res[:live]
        got_planes = (unpack_frag(Cp[0].view(torch.float16).float().cpu(), Mp, N).numpy().astype(np.float64)
                      + unpack_frag(Cp[1].view(torch.float16).float().cpu(), Mp, N).numpy() / X3_LO_SCALE)
        scale = np.abs(full).max()
        if epi == 1:
            got = Cc.cpu().numpy()
            assert np.isnan(got[live:]).all()                       # rows beyond the live count are not written
            assert np.abs(got[:live] - exact).max() < 5e-6 * scale, (N, K, np.abs(got[:live] - exact).max() / scale)
            assert np.abs(got[:live] - full).max() < 5e-6 * scale, (N, K, np.abs(got[:live] - full).max() / scale)   # 22-bit operands
            assert np.array_equal(unpack_frag32(Cp32.cpu(), Mp, N).numpy()[:live], got[:live])
            assert np.abs(got_planes[:live] - got[:live]).max() < 2e-6 * scale      # planes hold the row to 22 bits
            sq = ssq_out.cpu().numpy()
            want_sq = (got[:live].astype(np.float64).reshape(live, 48, 16) ** 2).sum(-1)
            assert np.abs(sq[:live] - want_sq).max() < 1e-5 * want_sq.max() and np.isnan(sq[live:]).all()
        else:
            assert np.abs(got_planes[:live] - exact).max() < 5e-6 * scale, (N, K, np.abs(got_planes[:live] - exact).max() / scale)
        assert np.isnan(got_planes[live:M]).all()


@pytest.mark.parametrize("force_mb", [0, 1, 2, 4])
@pytest.mark.parametrize("decode", [False, True, "tiled", "tiled128"])
def test_qkv_rope_fused(G, force_mb, decode):
    """perf-mode fused RMSNorm-scale + QKV + RoPE + KV append vs numpy (natural weight order); "tiled": a prompt-sized
    launch (M = 333 rows >= 256) goes to the LDS-tiled prefill kernel (prefill.hip) unless an M tile is forced"""
    from chattts_amd.engine import rope_row_perm, rope_tables
    lib = _lib.lib()
    rs = np.random.RandomState(17 + force_mb)
    if decode in ("tiled", "tiled128"):
        if force_mb not in (0, 4) or (decode == "tiled128" and force_mb):
            pytest.skip("one forced variant is enough at this size")
        B, T, cmax, nh, d = (9, 37, 60, 12, 64) if decode == "tiled" else (100, 37, 40, 12, 64)   # 333 / 3700 rows
        decode = False
    else:
        B, T, cmax, nh, d = (37, 1, 90, 12, 64) if decode else (3, 23, 60, 12, 64)
    M = B * T
    kv_start = rs.randint(0, 6, size=B).astype(np.int32)
    lens = (rs.randint(20, 60, size=B)).astype(np.int32)
    x32 = (rs.standard_normal((M, 768)) * 1.5).astype(f32)
    x_d = G.dev(x32)
    xb = torch.empty((M, 768), dtype=torch.bfloat16, device=G.DEV)
    ssq = torch.empty((M, 48), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_rows_prep(x_d.data_ptr(), xb.data_ptr(), ssq.data_ptr(), M, None), "rows_prep")
    W = G.bf16_round((rs.standard_normal((2304, 768)) * 0.05).astype(f32))
    perm = rope_row_perm().numpy()
    Wp = np.concatenate([W[:768][perm], W[768:1536][perm], W[1536:]], 0)
    cos, sin = rope_tables(128)
    keep = [G.dev(Wp, torch.bfloat16), G.dev(cos), G.dev(sin), G.dev(lens), G.dev(kv_start)]
    qkv = torch.zeros((M, 2304), dtype=torch.float32, device=G.DEV)
    kc = torch.zeros((B, nh, cmax, d), dtype=torch.bfloat16, device=G.DEV)
    vc = torch.zeros_like(kc)
    _lib.check(lib.ctts_k_qkv_rope(xb.data_ptr(), keep[0].data_ptr(), M, ssq.data_ptr(), 1e-6, qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                   cmax, keep[1].data_ptr(), keep[2].data_ptr(), 1 if decode else T, keep[3].data_ptr(), keep[4].data_ptr(),
                                   force_mb, None), "qkv_rope")
    torch.cuda.synchronize()
    rstd = 1.0 / np.sqrt((x32.astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
    y = (G.bf16_round(x32).astype(np.float64) @ W.astype(np.float64).T) * rstd
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, d, 2, dtype=np.float64) / d))).astype(f32)
    got_q = qkv.cpu().numpy()[:, :768]
    kcn, vcn = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    for m in range(M):
        b, slot = (m, lens[m] - 1) if decode else (m // T, m % T)
        pos = slot - kv_start[b]
        pos = 1 if pos < 0 else pos
        c, s_ = llama_np.rope_tables(np.array([pos]), inv_freq)

        def rope(v):
            v = v.reshape(nh, d)
            rot = np.concatenate([-v[:, d // 2:], v[:, : d // 2]], -1)
            return v * c[0] + rot * s_[0]
        rq, rk, rv = rope(y[m, :768]), rope(y[m, 768:1536]), y[m, 1536:].reshape(nh, d)
        assert np.abs(got_q[m].reshape(nh, d) - rq).max() < 2e-4 * max(1.0, np.abs(rq).max()), m
        assert np.abs(kcn[b, :, slot] - rk).max() < 1e-2 * max(1.0, np.abs(rk).max()), m
        assert np.abs(vcn[b, :, slot] - rv).max() < 1e-2 * max(1.0, np.abs(rv).max()), m


# asymmetric-B identity check: catches a row<->col swap in the C/D fragment mapping
def test_gemm_skinny_identity(G):
    K = 768
    A = np.zeros((16, K), f32)
    A[np.arange(16), np.arange(16)] = 1.0
    W = (np.arange(32 * K, dtype=f32).reshape(32, K) % 97) * 0.25
    for wt in ("f32", "bf16"):
        got = G.gemm(A, W, wt=wt)
        assert np.array_equal(got, W[:, :16].T), wt


# ------------------------------------------------------------------------------------------------
# tiled GEMM (codec dense layers incl. conv-as-GEMM)
# ------------------------------------------------------------------------------------------------
def _conv_ref(X, Wt, B, F, taps, pad):
    """X [B*F, Cin], Wt [Cout, Cin, k] torch layout -> [B*F, Cout] float64"""
    Cin = X.shape[1]
    x = X.reshape(B, F, Cin)
    y = codec_np.conv1d_cl(x, Wt.astype(f32), None, pad=pad).astype(np.float64)
    return y.reshape(B * F, -1)


@pytest.mark.parametrize("case", [
    dict(B=3, F=50, cin=384, cout=128, taps=3, pad=1, epi=4),    # conv_in.0 + bias + GELU
    dict(B=3, F=50, cin=128, cout=512, taps=3, pad=1, epi=3),    # conv_in.2 + bias
    dict(B=2, F=37, cin=100, cout=512, taps=7, pad=3, epi=3),    # vocos embed (K = 700, ragged)
    dict(B=2, F=37, cin=384, cout=100, taps=3, pad=1, epi=6),    # out_conv * coef (N = 100)
    dict(B=1, F=5, cin=384, cout=128, taps=3, pad=1, epi=4),     # fewer frames than one tile
])
@pytest.mark.parametrize("tiled", [1, 2])
def test_gemm_tiled_conv(G, case, tiled):
    c = case
    rs = np.random.RandomState(c["cin"] + c["cout"])
    X = rs.standard_normal((c["B"] * c["F"], c["cin"])).astype(f32)
    Wt = (rs.standard_normal((c["cout"], c["cin"], c["taps"])) / np.sqrt(c["cin"] * c["taps"])).astype(f32)
    Wp = np.ascontiguousarray(Wt.transpose(0, 2, 1)).reshape(c["cout"], c["taps"] * c["cin"])
    bias = rs.standard_normal(c["cout"]).astype(f32) * 0.1
    gam = (0.5 + rs.rand(c["cout"])).astype(f32)
    got = G.gemm(X, Wp, tiled=tiled, epi=c["epi"], bias=bias, gamma=gam, taps=c["taps"], cin=c["cin"], frames=c["F"], pad=c["pad"])
    acc = _conv_ref(X, Wt, c["B"], c["F"], c["taps"], c["pad"])
    if c["epi"] == 4:
        ref = codec_np.gelu((acc + bias).astype(f32))
    elif c["epi"] == 3:
        ref = acc + bias
    else:
        ref = acc * gam
    assert G.relerr(got, ref) < (1e-5 if tiled == 1 else 3e-5), G.relerr(got, ref)  # split-bf16: 2^-16-class products


@pytest.mark.parametrize("tiled", [1, 2])
@pytest.mark.parametrize("M,N,K,epi", [(200, 2048, 512, 4), (200, 512, 2048, 5), (130, 384, 512, 0), (70, 1026, 512, 3), (64, 1536, 512, 4),
                                       (300, 100, 1152, 6), (2304, 512, 2048, 5), (2100, 1536, 512, 4)])
def test_gemm_tiled_linear(G, M, N, K, epi, tiled):
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((M, K)).astype(f32)
    W = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(f32)
    bias = rs.standard_normal(N).astype(f32) * 0.1
    gam = (0.05 + 0.1 * rs.rand(N)).astype(f32)
    res = rs.standard_normal((M, N)).astype(f32)
    got = G.gemm(A, W, tiled=tiled, epi=epi, bias=bias, gamma=gam, res=res)
    acc = A.astype(np.float64) @ W.astype(np.float64).T
    ref = {0: acc, 3: acc + bias, 4: codec_np.gelu((acc + bias).astype(f32)), 5: res + gam * (acc + bias), 6: acc * gam}[epi]
    assert G.relerr(got, ref) < (1e-5 if tiled == 1 else 3e-5), G.relerr(got, ref)


# the 256x256 split-bf16 tile with two LDS buffers is what the acoustic decoder of a BASELINE-size batch runs (selected from
# M >= 12288 rows, gemm.hip launch_gemm_tiled_bf16x3): every layer shape of DVAE / Vocos at that size, vs float64
@pytest.mark.parametrize("M", [12288, 16400])
@pytest.mark.parametrize("N,K,epi", [(2048, 512, 4), (512, 2048, 5), (1536, 512, 4), (512, 1536, 5), (1026, 512, 3)])
def test_gemm_tiled_bf16x3_big_tile_linear(G, M, N, K, epi):
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((M, K)).astype(f32)
    W = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(f32)
    bias = rs.standard_normal(N).astype(f32) * 0.1
    gam = (0.05 + 0.1 * rs.rand(N)).astype(f32)
    res = rs.standard_normal((M, N)).astype(f32)
    got = G.gemm(A, W, tiled=2, epi=epi, bias=bias, gamma=gam, res=res)
    acc = A.astype(np.float64) @ W.astype(np.float64).T
    ref = {3: acc + bias, 4: codec_np.gelu((acc + bias).astype(f32)), 5: res + gam * (acc + bias)}[epi]
    assert np.isfinite(got).all()
    assert G.relerr(got, ref) < 3e-5, G.relerr(got, ref)


@pytest.mark.parametrize("M", [12288, 16400, 300])
@pytest.mark.parametrize("N,K,epi", [(2048, 512, 0), (512, 2048, 1), (1536, 512, 0), (512, 1536, 1)])
def test_gemm_h1p(G, M, N, K, epi):
    """the same pair on ONE fp16 plane per operand (gemm_mode 2, csrc/codec_gemm.hip: gemm_h1p_k).  Against float64 on the SAME
    fp16-rounded operands the kernel is exact up to f32 accumulation (3e-5, the x3p bar); against the unrounded operands the
    stated bound is the rounding of both operands to 11 significant bits: relative error of the result < 1e-3.  The GELU
    epilogue's output is itself rounded to fp16 (one more 2^-11)."""
    from chattts_amd.engine import pack_h1p, unpack_h1p
    lib = _lib.lib()
    rs = np.random.RandomState(M + N + K + 1)
    Mp = (M + 255) // 256 * 256
    A = np.zeros((Mp, K), f32)
    A[:M] = rs.standard_normal((M, K)).astype(f32)
    W = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(f32)
    bias = rs.standard_normal(N).astype(f32) * 0.1
    if epi == 0:
        bias[0] = 1e5                               # column 0 of gelu(.) leaves the half range: saturated to 65504 where it is rounded, never inf
    gam = (0.05 + 0.1 * rs.rand(N)).astype(f32)
    res = rs.standard_normal((M, N)).astype(f32)
    Ap_h = pack_h1p(torch.from_numpy(A))
    Wp_h = pack_h1p(torch.from_numpy(W))
    Ar, Wr = unpack_h1p(Ap_h, Mp, K).numpy(), unpack_h1p(Wp_h, N, K).numpy()      # what the kernel multiplies
    Ap, Wp = Ap_h.to(G.DEV), Wp_h.to(G.DEV)
    b_d, g_d = G.dev(bias), G.dev(gam)
    acc = Ar[:M].astype(np.float64) @ Wr.astype(np.float64).T
    acc_full = A[:M].astype(np.float64) @ W.astype(np.float64).T
    if epi == 0:
        Cp = torch.zeros((Mp * N,), dtype=torch.float16, device=G.DEV)
        _lib.check(lib.ctts_k_gemm_h1p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 0, b_d.data_ptr(), None, None, None, Cp.data_ptr(), None), "h1p gelu")
        torch.cuda.synchronize()
        got = unpack_h1p(Cp.cpu(), Mp, N).numpy()[:M]
        assert (got[:, 0] == 65504.0).all()
        got = got[:, 1:]
        ref = codec_np.gelu((acc + bias).astype(f32))[:, 1:]
        ref_full = codec_np.gelu((acc_full + bias).astype(f32))[:, 1:]
        tol_same = 6e-4                              # the output's own fp16 rounding
    else:
        C_d = G.dev(res).clone()
        _lib.check(lib.ctts_k_gemm_h1p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 1, b_d.data_ptr(), g_d.data_ptr(), C_d.data_ptr(), C_d.data_ptr(),
                                       None, None), "h1p res")
        torch.cuda.synchronize()
        got = C_d.cpu().numpy()
        ref = res + gam * (acc + bias)
        ref_full = res + gam * (acc_full + bias)
        tol_same = 3e-5
    assert np.isfinite(got).all()
    assert G.relerr(got, ref) < tol_same, G.relerr(got, ref)
    assert G.relerr(got, ref_full) < 1e-3, G.relerr(got, ref_full)


@pytest.mark.parametrize("M", [12288, 16400, 300])
@pytest.mark.parametrize("N,K,epi", [(2048, 512, 0), (512, 2048, 1), (1536, 512, 0), (512, 1536, 1)])
def test_gemm_x3p(G, M, N, K, epi):
    """split-bf16 GEMM on pre-split fragment-order planes, LDS-DMA staged (csrc/codec_gemm.hip): both epilogues of the ConvNeXt
    point-wise pair at every DVAE / Vocos shape, ragged last row tile (M not a multiple of 256), vs float64"""
    from chattts_amd.engine import pack_x3p, unpack_x3p
    lib = _lib.lib()
    rs = np.random.RandomState(M + N + K)
    Mp = (M + 255) // 256 * 256
    A = np.zeros((Mp, K), f32)
    A[:M] = rs.standard_normal((M, K)).astype(f32)
    W = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(f32)
    bias = rs.standard_normal(N).astype(f32) * 0.1
    gam = (0.05 + 0.1 * rs.rand(N)).astype(f32)
    res = rs.standard_normal((M, N)).astype(f32)
    Ap = pack_x3p(torch.from_numpy(A)).to(G.DEV)
    Wp = pack_x3p(torch.from_numpy(W)).to(G.DEV)
    b_d, g_d = G.dev(bias), G.dev(gam)
    acc = A[:M].astype(np.float64) @ W.astype(np.float64).T
    if epi == 0:
        Cp = torch.zeros((Mp * N * 2,), dtype=torch.bfloat16, device=G.DEV)
        _lib.check(lib.ctts_k_gemm_x3p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 0, b_d.data_ptr(), None, None, None, Cp.data_ptr(), None), "x3p gelu")
        torch.cuda.synchronize()
        got = unpack_x3p(Cp.cpu(), Mp, N).numpy()[:M]
        ref = codec_np.gelu((acc + bias).astype(f32))
    else:
        C_d = G.dev(res).clone()
        _lib.check(lib.ctts_k_gemm_x3p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 1, b_d.data_ptr(), g_d.data_ptr(), C_d.data_ptr(), C_d.data_ptr(),
                                       None, None), "x3p res")
        torch.cuda.synchronize()
        got = C_d.cpu().numpy()
        ref = res + gam * (acc + bias)
    assert np.isfinite(got).all()
    assert G.relerr(got, ref) < 3e-5, G.relerr(got, ref)


@pytest.mark.parametrize("kind", ["x3p", "h1p"])
@pytest.mark.parametrize("M", [16400, 300, 100, 129])
@pytest.mark.parametrize("N,K,epi", [(1536, 512, 0), (512, 1536, 1), (1024, 256, 0), (256, 1024, 1), (256, 64, 0), (256, 64, 1), (512, 128, 1), (256, 192, 0)])
def test_codec_gemm_tilings_are_bit_identical(G, monkeypatch, kind, M, N, K, epi):
    """round 6: the 128 x 256 tiles (two workgroups of four waves per CU: one's epilogue under the other's MFMAs) issue the same MFMAs in
    the same k order per accumulator and run the same epilogues as the 256 x 256 tiles (CTTS_CODEC_TILE=256) -- every output bit equal,
    whole and ragged row tiles (the last 128-row tile empty, half full, one row), both epilogues, both operand formats; K down to 64 (one
    or two ring slots: the prologue's short paths)."""
    from chattts_amd.engine import pack_h1p, pack_x3p
    lib = _lib.lib()
    rs = np.random.RandomState(M + N + K + 7)
    Mp = (M + 255) // 256 * 256
    A = np.zeros((Mp, K), f32)
    A[:M] = rs.standard_normal((M, K)).astype(f32)
    W = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(f32)
    pack, fn, planes, dt = (pack_x3p, lib.ctts_k_gemm_x3p, 2, torch.bfloat16) if kind == "x3p" else (pack_h1p, lib.ctts_k_gemm_h1p, 1, torch.float16)
    Ap, Wp = pack(torch.from_numpy(A)).to(G.DEV), pack(torch.from_numpy(W)).to(G.DEV)
    b_d, g_d = G.dev(rs.standard_normal(N).astype(f32) * 0.1), G.dev((0.05 + 0.1 * rs.rand(N)).astype(f32))
    res = rs.standard_normal((M, N)).astype(f32)
    outs = []
    for tile in ("256", "128"):
        monkeypatch.setenv("CTTS_CODEC_TILE", tile)
        if epi == 0:
            Cp = torch.zeros((Mp * N * planes,), dtype=dt, device=G.DEV)
            _lib.check(fn(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 0, b_d.data_ptr(), None, None, None, Cp.data_ptr(), None), kind)
            torch.cuda.synchronize()
            full = Cp.view(torch.int16).cpu().numpy()
            # rows >= M of the last 256-row tile are padding: the 128-row tiling does not write the second half when it holds no row
            if kind == "x3p":
                keep = full.reshape(Mp // 32, N // 16, 2, 2, 32, 8)[:, :, :, :, :, :]
                rows = (np.arange(Mp // 32)[:, None] * 32 + np.arange(32)[None, :])            # [tile, r]
                outs.append(keep[np.broadcast_to((rows < M)[:, None, None, None, :, None], keep.shape)])
            else:
                keep = full.reshape(Mp // 32, N // 16, 2, 32, 8)
                rows = (np.arange(Mp // 32)[:, None] * 32 + np.arange(32)[None, :])
                outs.append(keep[np.broadcast_to((rows < M)[:, None, None, :, None], keep.shape)])
        else:
            C_d = G.dev(res).clone()
            _lib.check(fn(Ap.data_ptr(), Wp.data_ptr(), M, N, K, 1, b_d.data_ptr(), g_d.data_ptr(), C_d.data_ptr(), C_d.data_ptr(), None, None), kind)
            torch.cuda.synchronize()
            outs.append(C_d.cpu().numpy().view(np.int32))
    assert outs[0].size > 0 and np.array_equal(outs[0], outs[1])

@pytest.mark.parametrize("M", [32768 + 128, 1000, 128, 77])
@pytest.mark.parametrize("inter", [1536, 2048, 256])
def test_mlp_fused_equals_the_two_launches(G, M, inter):
    """round 6: one ConvNeXt MLP (pwconv1 -> GELU -> pwconv2 -> gamma -> residual, dvae.py:46-66) in ONE launch, the inter-wide activation
    kept on the CU (csrc/codec_gemm.hip mlp_fused_h1p_k) -- every output bit equal to ctts_k_gemm_h1p(GELU_PACKED) followed by
    ctts_k_gemm_h1p(SCALE_RES): whole and ragged 128-row tiles, more tiles than CUs, both widths of the decoder (Vocos 1536, DVAE 2048) and a
    single chunk; and the pair itself against float64 within the fp16 plane's bound."""
    from chattts_amd.engine import pack_h1p
    lib = _lib.lib()
    rs = np.random.RandomState(M + inter)
    Mp = (M + 255) // 256 * 256
    A = np.zeros((Mp, 512), f32)
    A[:M] = rs.standard_normal((M, 512)).astype(f32)
    W1 = (rs.standard_normal((inter, 512)) / np.sqrt(512)).astype(f32)
    W2 = (rs.standard_normal((512, inter)) / np.sqrt(inter)).astype(f32)
    b1, b2 = rs.standard_normal(inter).astype(f32) * 0.1, rs.standard_normal(512).astype(f32) * 0.1
    gam = (0.05 + 0.1 * rs.rand(512)).astype(f32)
    res = rs.standard_normal((M, 512)).astype(f32)
    Ap, W1p, W2p = (pack_h1p(torch.from_numpy(x)).to(G.DEV) for x in (A, W1, W2))
    b1_d, b2_d, g_d = G.dev(b1), G.dev(b2), G.dev(gam)
    # the two launches
    Hp = torch.zeros((Mp * inter,), dtype=torch.float16, device=G.DEV)
    C2 = G.dev(res).clone()
    _lib.check(lib.ctts_k_gemm_h1p(Ap.data_ptr(), W1p.data_ptr(), M, inter, 512, 0, b1_d.data_ptr(), None, None, None, Hp.data_ptr(), None), "pwconv1")
    _lib.check(lib.ctts_k_gemm_h1p(Hp.data_ptr(), W2p.data_ptr(), M, 512, inter, 1, b2_d.data_ptr(), g_d.data_ptr(), C2.data_ptr(), C2.data_ptr(), None, None), "pwconv2")
    # one launch
    C1 = G.dev(res).clone()
    _lib.check(lib.ctts_k_mlp_fused(Ap.data_ptr(), W1p.data_ptr(), W2p.data_ptr(), M, inter, b1_d.data_ptr(), b2_d.data_ptr(), g_d.data_ptr(),
                                    C1.data_ptr(), 1, None), "mlp_fused")
    torch.cuda.synchronize()
    got, two = C1.cpu().numpy(), C2.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got.view(np.int32), two.view(np.int32)), float(np.abs(got - two).max())
    from scipy.special import erf
    h = A[:M].astype(np.float64) @ W1.astype(np.float64).T + b1
    h = 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
    ref = res + gam * (h @ W2.astype(np.float64).T + b2)
    assert G.relerr(got, ref) < 2e-4, G.relerr(got, ref)


def test_gemm_tiled_bf16x3_big_tile_conv(G):
    """conv-as-GEMM gather (taps 3, zero padding at both utterance ends) on the two-buffer 256x256 tile: conv_in.2 of the
    DVAE decoder at 4 x 3100 frames (M = 12400 >= 12288)"""
    B, F, cin, cout, taps, pad = 4, 3100, 128, 512, 3, 1
    rs = np.random.RandomState(17)
    X = rs.standard_normal((B * F, cin)).astype(f32)
    Wt = (rs.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(f32)
    Wp = np.ascontiguousarray(Wt.transpose(0, 2, 1)).reshape(cout, taps * cin)
    bias = rs.standard_normal(cout).astype(f32) * 0.1
    got = G.gemm(X, Wp, tiled=2, epi=3, bias=bias, taps=taps, cin=cin, frames=F, pad=pad)
    x = X.reshape(B, F, cin).astype(np.float64)
    xp = np.zeros((B, F + 2 * pad, cin))
    xp[:, pad: pad + F] = x
    ref = sum(xp[:, j: j + F] @ Wt[:, :, j].astype(np.float64).T for j in range(taps)).reshape(B * F, cout) + bias
    assert G.relerr(got, ref) < 3e-5, G.relerr(got, ref)


# ------------------------------------------------------------------------------------------------
# RoPE + KV append + attention (prefill rows, then a decode row), both KV dtypes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kv", ["f32", "bf16"])
def test_rope_attention(G, kv):
    from chattts_amd.engine import rope_tables
    lib = _lib.lib()
    B, T, nh, d, H = 3, 21, 12, 64, 768
    cmax = T + 80
    kv_start = np.array([0, 4, 13], np.int32)
    rs = np.random.RandomState(5)
    code = _lib.BF16 if kv == "bf16" else _lib.F32
    tdt = torch.bfloat16 if kv == "bf16" else torch.float32
    cos, sin = rope_tables(256)
    cos_d, sin_d = G.dev(cos), G.dev(sin)
    kc = torch.zeros((B, nh, cmax, d), dtype=tdt, device=G.DEV)
    vc = torch.zeros_like(kc)
    ks_d = G.dev(kv_start)
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, d, 2, dtype=np.float64) / d))).astype(f32)

    def ref_rows(qkv, slots_b, Kc, Vc):
        """numpy attention for query rows given as list of (b, slot); Kc/Vc [B,nh,cmax,d] f32 (already appended)"""
        out = np.zeros((len(slots_b), H), f32)
        for r, (b, slot) in enumerate(slots_b):
            lo = min(kv_start[b], slot)
            q = qkv[r, :H].reshape(nh, d)
            Kb, Vb = Kc[b, :, lo: slot + 1], Vc[b, :, lo: slot + 1]
            s = np.einsum("hd,hjd->hj", q.astype(np.float64), Kb.astype(np.float64)) * 0.125
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            out[r] = np.einsum("hj,hjd->hd", p, Vb.astype(np.float64)).reshape(H)
        return out

    def rope_np(x, pos):  # x [nh, d]
        c, s_ = llama_np.rope_tables(np.array([pos]), inv_freq)
        rot = np.concatenate([-x[:, d // 2:], x[:, : d // 2]], -1)
        return x * c[0] + rot * s_[0]

    Kc = np.zeros((B, nh, cmax, d), f32)
    Vc = np.zeros_like(Kc)
    # ---- prefill: M = B*T rows
    qkv = rs.standard_normal((B * T, 3 * H)).astype(f32)
    qkv_d = G.dev(qkv)
    _lib.check(lib.ctts_k_rope_append(qkv_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), code, cmax, cos_d.data_ptr(), sin_d.data_ptr(),
                                      T, None, ks_d.data_ptr(), B * T, None), "rope")
    out_d = torch.full((B * T, H), float("nan"), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_attention(qkv_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), code, cmax, out_d.data_ptr(), T, None,
                                    ks_d.data_ptr(), B * T, None), "attn")
    torch.cuda.synchronize()
    qkv_r = qkv.copy()
    rows = []
    for b in range(B):
        for t in range(T):
            pos = t - kv_start[b]
            pos = 1 if pos < 0 else pos
            m = b * T + t
            qkv_r[m, :H] = rope_np(qkv[m, :H].reshape(nh, d), pos).reshape(H)
            Kc[b, :, t] = rope_np(qkv[m, H:2 * H].reshape(nh, d), pos)
            Vc[b, :, t] = qkv[m, 2 * H:].reshape(nh, d)
            rows.append((b, t))
    if kv == "bf16":
        Kc, Vc = G.bf16_round(Kc), G.bf16_round(Vc)
    got_q = qkv_d.cpu().numpy()[:, :H]
    assert np.abs(got_q - qkv_r[:, :H]).max() < 1e-5
    assert np.abs(kc.float().cpu().numpy()[:, :, :T] - Kc[:, :, :T]).max() < (1e-5 if kv == "f32" else 1e-2)
    ref = ref_rows(qkv_r, rows, Kc, Vc)
    got = out_d.cpu().numpy()
    valid = np.array([t >= kv_start[b] for b, t in rows])
    assert np.isfinite(got).all()
    assert np.abs(got[valid] - ref[valid]).max() < 2e-5, np.abs(got[valid] - ref[valid]).max()
    # ---- decode rows at different context lengths (exercise multi-block / multi-wave paths)
    for extra in (1, 37, 70):
        # fill the cache with random K/V up to len-1, then append one row per b
        lens = np.array([T + extra, T + extra, T + extra], np.int32)
        fillK = rs.standard_normal((B, nh, cmax, d)).astype(f32)
        fillV = rs.standard_normal((B, nh, cmax, d)).astype(f32)
        if kv == "bf16":
            fillK, fillV = G.bf16_round(fillK), G.bf16_round(fillV)
        kc.copy_(G.dev(fillK).to(tdt))
        vc.copy_(G.dev(fillV).to(tdt))
        qkv1 = rs.standard_normal((B, 3 * H)).astype(f32)
        q1_d = G.dev(qkv1)
        len_d = G.dev(lens)
        _lib.check(lib.ctts_k_rope_append(q1_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), code, cmax, cos_d.data_ptr(), sin_d.data_ptr(),
                                          1, len_d.data_ptr(), ks_d.data_ptr(), B, None), "rope1")
        o1 = torch.full((B, H), float("nan"), dtype=torch.float32, device=G.DEV)
        _lib.check(lib.ctts_k_attention(q1_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), code, cmax, o1.data_ptr(), 1, len_d.data_ptr(),
                                        ks_d.data_ptr(), B, None), "attn1")
        torch.cuda.synchronize()
        Kr, Vr = fillK.copy(), fillV.copy()
        qr = qkv1.copy()
        rows1 = []
        for b in range(B):
            slot = lens[b] - 1
            pos = slot - kv_start[b]
            qr[b, :H] = rope_np(qkv1[b, :H].reshape(nh, d), pos).reshape(H)
            Kr[b, :, slot] = rope_np(qkv1[b, H:2 * H].reshape(nh, d), pos)
            Vr[b, :, slot] = qkv1[b, 2 * H:].reshape(nh, d)
            rows1.append((b, slot))
        if kv == "bf16":
            Kr, Vr = G.bf16_round(Kr), G.bf16_round(Vr)
        ref1 = ref_rows(qr, rows1, Kr, Vr)
        err = np.abs(o1.cpu().numpy() - ref1).max()
        assert err < 2e-5, (extra, err)


@pytest.mark.parametrize("T,slot0", [(128, 0), (300, 0), (512, 0), (200, 312)])
def test_attention_prefill_mfma(G, T, slot0):
    """flash-style MFMA prefill attention (perf mode, prompt chunks >= 128 rows): causal + left-pad mask, ragged last query tile,
    a later chunk of a longer prompt (slot0 > 0: the first 312 keys come from the cache), vs float64.  P enters the second
    product as bf16 (like every activation of the perf mode): tolerance 1e-2 on outputs of magnitude ~1."""
    lib = _lib.lib()
    rs = np.random.RandomState(T + slot0)
    B, nh, d, H = 3, 12, 64, 768
    cmax = slot0 + T + 40
    kv_start = np.array([0, 7, min(150, slot0 + T - 3)], np.int32)
    Kc = G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32))
    Vc = G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32))
    kc, vc = G.dev(Kc, torch.bfloat16), G.dev(Vc, torch.bfloat16)
    # rows of the cache the kernel must never use: poison them (beyond the chunk's last slot)
    kc[:, :, slot0 + T:] = float("nan")
    vc[:, :, slot0 + T:] = float("nan")
    qkv = rs.standard_normal((B * T, 3 * H)).astype(f32)
    q_d, ks_d = G.dev(qkv), G.dev(kv_start)
    out_d = torch.full((B * T, H), float("nan"), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_attention_prefill(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), cmax, out_d.data_ptr(), T, slot0, ks_d.data_ptr(),
                                            B * T, None), "attn prefill")
    torch.cuda.synchronize()
    got = out_d.cpu().numpy()
    assert np.isfinite(got).all()
    worst = 0.0
    for b in range(B):
        for t in sorted(set(rs.randint(0, T, size=24).tolist() + [0, T - 1])):
            slot = slot0 + t
            lo = min(kv_start[b], slot)
            q = qkv[b * T + t, :H].reshape(nh, d).astype(np.float64)
            Kb, Vb = Kc[b, :, lo: slot + 1].astype(np.float64), Vc[b, :, lo: slot + 1].astype(np.float64)
            sc = np.einsum("hd,hjd->hj", q, Kb) * 0.125
            p = np.exp(sc - sc.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            ref = np.einsum("hj,hjd->hd", p, Vb).reshape(H)
            if slot >= kv_start[b]:      # pad query rows see only themselves; their output is never consumed
                worst = max(worst, float(np.abs(got[b * T + t] - ref).max()))
    assert worst < 1e-2, worst


@pytest.mark.parametrize("n_live", [1, 3, 10, 21, 22, 40, 45, 64])
def test_attention_decode_remainder_split(G, n_live):
    """decode attention of the perf mode (bf16 KV, packed bf16 output) with remainder splitting: 12 * n_live units on 256 CUs
    -- whole units, units cut into 2..8 key ranges that meet through memory (write-through partials, ticket counter, last
    arriver merges) -- equals the float64 reference and, to bf16 rounding, the unsplit kernel; three launches in a row
    re-use the partial buffers and the counters (each last arriver resets its own)."""
    from chattts_amd.engine import unpack_frag
    lib = _lib.lib()
    rs = np.random.RandomState(100 + n_live)
    B, nh, d, H, cmax = 64, 12, 64, 768, 640
    Bp = 64
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    kc = G.dev(G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32)), torch.bfloat16)
    vc = G.dev(G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32)), torch.bfloat16)
    Kc, Vc = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    part = torch.zeros((512 * 8 * 66,), dtype=torch.float32, device=G.DEV)
    cnt = torch.zeros((512,), dtype=torch.int32, device=G.DEV)
    na = G.dev(np.array([n_live], np.int32))
    for rep in range(3):
        slots_b = rs.permutation(B)[:n_live]                       # compact row m -> utterance slot
        jlo = rs.randint(0, 30, size=n_live)
        slot = np.array([rs.randint(jlo[m] + 1, cmax) if rs.rand() < 0.8 else jlo[m] + rs.randint(0, 12) for m in range(n_live)])
        desc = np.zeros((Bp, 4), np.int32)
        desc[:n_live, 0], desc[:n_live, 1], desc[:n_live, 2], desc[:n_live, 3] = slots_b, slot, slot - jlo, jlo
        qkv = rs.standard_normal((Bp, 3 * H)).astype(f32)
        q_d, desc_d = G.dev(qkv), G.dev(desc)
        outs = []
        for n_cu in (ncu, 0):
            o = torch.full((Bp * H,), float("nan"), dtype=torch.bfloat16, device=G.DEV)
            _lib.check(lib.ctts_k_attention_dec(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), cmax, o.data_ptr(), desc_d.data_ptr(),
                                                na.data_ptr(), Bp, part.data_ptr(), cnt.data_ptr(), n_cu, None), "attention_dec")
            torch.cuda.synchronize()
            outs.append(unpack_frag(o.float().cpu(), Bp, H).numpy())
        assert int(cnt.abs().sum()) == 0                            # every counter went back to zero
        ref = np.zeros((n_live, H))
        for m in range(n_live):
            b = slots_b[m]
            q = qkv[m, :H].reshape(nh, d).astype(np.float64)
            Kb, Vb = Kc[b, :, jlo[m]: slot[m] + 1].astype(np.float64), Vc[b, :, jlo[m]: slot[m] + 1].astype(np.float64)
            sc = np.einsum("hd,hjd->hj", q, Kb) * 0.125
            p = np.exp(sc - sc.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            ref[m] = np.einsum("hj,hjd->hd", p, Vb).reshape(H)
        split, whole = outs
        assert np.isfinite(split[:n_live]).all() and np.isnan(split[n_live:]).all()
        assert np.abs(split[:n_live] - ref).max() < 1.5e-2, np.abs(split[:n_live] - ref).max()      # bf16 output (|o| <~ 3)
        assert np.abs(split[:n_live] - whole[:n_live]).max() < 1.6e-2                                # at most one bf16 ulp apart
        assert (split[:n_live] != whole[:n_live]).mean() < 0.02


@pytest.mark.parametrize("mode", ["bf16", "f32"])
@pytest.mark.parametrize("n_live", [1, 5, 22, 43, 64])
def test_attention_decode_persistent_grid(G, mode, n_live):
    """round 5: the decode step's attention on a persistent grid (attention_persist_k: G workgroups walk the live (utterance, head)
    units in snake order, a ring of 2 / 3 / 4 KV blocks per wave that runs on into the workgroup's next unit) against the float64
    softmax(q K^T / 8 + mask) V reference (examples/onnx/modeling_llama.py:455-475), and BIT-IDENTICAL to one workgroup per unit
    (attention_k) for every grid size and ring depth -- descriptors that cover all rows (absent rows b = -1, as the decode step writes
    them), a host-compacted batch (n_active bounds the list, stale descriptors behind it), a finished row in the middle of the list,
    contexts of 1 key, rows that must stay untouched."""
    from chattts_amd.engine import unpack_frag, unpack_frag32
    lib = _lib.lib()
    rs = np.random.RandomState(500 + n_live + (7 if mode == "f32" else 0))
    B, nh, d, H, cmax, Bp = 64, 12, 64, 768, 560, 64
    bf = mode == "bf16"
    kv_t = torch.bfloat16 if bf else torch.float32
    Kc = rs.standard_normal((B, nh, cmax, d)).astype(f32)
    Vc = rs.standard_normal((B, nh, cmax, d)).astype(f32)
    if bf:
        Kc, Vc = G.bf16_round(Kc), G.bf16_round(Vc)
    kc, vc = G.dev(Kc, kv_t), G.dev(Vc, kv_t)
    unpack = unpack_frag if bf else unpack_frag32
    try:
        for rep, covers_all in enumerate((1, 0, 1)):
            slots_b = rs.permutation(B)[:n_live]
            jlo = rs.randint(0, 30, size=n_live)
            slot = np.array([rs.randint(jlo[m] + 1, cmax) if rs.rand() < 0.8 else jlo[m] + rs.randint(0, 12) for m in range(n_live)])
            order = np.argsort(-(slot - jlo), kind="stable")               # descending context, as the decode step orders its rows
            slots_b, jlo, slot = slots_b[order], jlo[order], slot[order]
            desc = np.zeros((Bp, 4), np.int32)
            desc[:, 0] = -1 if covers_all else rs.randint(0, B, size=Bp)    # host-compacted: stale but plausible rows behind n_active
            desc[:, 1] = rs.randint(0, cmax, size=Bp)
            desc[:n_live, 0], desc[:n_live, 1], desc[:n_live, 2], desc[:n_live, 3] = slots_b, slot, slot - jlo, jlo
            dead = -1
            if rep == 2 and n_live >= 5:                                    # a row that finished since the last compaction
                dead = n_live // 2
                desc[dead, 0] = -1
            qkv = rs.standard_normal((Bp, 3 * H)).astype(f32)
            q_d, desc_d = G.dev(qkv), G.dev(desc)
            na = G.dev(np.array([n_live], np.int32))

            def run(persist, g, ring):
                _lib.check(lib.ctts_k_attention_cfg(persist, g, ring), "attention_cfg")
                o = torch.full((Bp * H,), float("nan"), dtype=kv_t, device=G.DEV)
                _lib.check(lib.ctts_k_attention_dec2(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), _lib.BF16 if bf else _lib.F32, cmax,
                                                     o.data_ptr(), desc_d.data_ptr(), None if covers_all else na.data_ptr(), covers_all,
                                                     Bp, None), "attention_dec2")
                torch.cuda.synchronize()
                return unpack(o.float().cpu(), Bp, H).numpy()

            unit = run(0, 0, 0)
            live = [m for m in range(n_live) if m != dead]
            if not bf:     # the f32x3 parity mode's output: the SAME f32 result, stored as split-fp16 planes (hi = fp16(o), lo' = fp16((o - hi) 2^11))
                from chattts_amd.engine import split_f16
                for persist in (0, 1):
                    _lib.check(lib.ctts_k_attention_cfg(persist, 256, 4), "attention_cfg")
                    pl = torch.full((2, Bp * H), float("nan"), dtype=torch.float32, device=G.DEV).to(torch.bfloat16)
                    _lib.check(lib.ctts_k_attention_dec2(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), 2, cmax, pl.data_ptr(), desc_d.data_ptr(),
                                                         None if covers_all else na.data_ptr(), covers_all, Bp, None), "attention_dec2 planes")
                    torch.cuda.synchronize()
                    hi = unpack_frag(pl[0].view(torch.float16).cpu(), Bp, H)
                    lo = unpack_frag(pl[1].view(torch.float16).cpu(), Bp, H)
                    want_hi, want_lo = split_f16(torch.from_numpy(unit))
                    assert torch.equal(hi[live], want_hi[live]) and torch.equal(lo[live], want_lo[live]), persist
            ref = np.zeros((Bp, H))
            for m in live:
                b = slots_b[m]
                q = qkv[m, :H].reshape(nh, d).astype(np.float64)
                Kb, Vb = Kc[b, :, jlo[m]: slot[m] + 1].astype(np.float64), Vc[b, :, jlo[m]: slot[m] + 1].astype(np.float64)
                sc = np.einsum("hd,hjd->hj", q, Kb) * 0.125
                pr = np.exp(sc - sc.max(-1, keepdims=True))
                pr /= pr.sum(-1, keepdims=True)
                ref[m] = np.einsum("hj,hjd->hd", pr, Vb).reshape(H)
            untouched = [m for m in range(Bp) if m not in live]
            assert np.isfinite(unit[live]).all() and np.isnan(unit[untouched]).all()
            assert np.abs(unit[live] - ref[live]).max() < (1.5e-2 if bf else 2e-5)
            for g, ring in ((256, 4), (256, 2), (256, 3), (7, 4), (100, 3), (1, 2), (768, 4), (331, 4)):
                got = run(1, g, ring)
                assert np.isnan(got[untouched]).all(), (g, ring)
                assert np.array_equal(got[live], unit[live]), (g, ring, np.abs(got[live] - unit[live]).max())
            # 2 / 3 / 4 heads of one utterance per workgroup (attention_hpw_k): the same units, the same bits, a third / a quarter of the workgroups
            for hpw in (2, 3, 4):
                _lib.check(lib.ctts_k_attention_heads_per_wg(hpw), "heads_per_wg")
                got = run(0, 0, 0)
                assert np.isnan(got[untouched]).all(), hpw
                assert np.array_equal(got[live], unit[live]), (hpw, np.abs(got[live] - unit[live]).max())
                if not bf:
                    pl = torch.full((2, Bp * H), float("nan"), dtype=torch.float32, device=G.DEV).to(torch.bfloat16)
                    _lib.check(lib.ctts_k_attention_dec2(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), 2, cmax, pl.data_ptr(), desc_d.data_ptr(),
                                                         None if covers_all else na.data_ptr(), covers_all, Bp, None), "attention_dec2 planes")
                    torch.cuda.synchronize()
                    want_hi, want_lo = split_f16(torch.from_numpy(unit))
                    assert torch.equal(unpack_frag(pl[0].view(torch.float16).cpu(), Bp, H)[live], want_hi[live])
                    assert torch.equal(unpack_frag(pl[1].view(torch.float16).cpu(), Bp, H)[live], want_lo[live])
            _lib.check(lib.ctts_k_attention_heads_per_wg(1), "heads_per_wg")
    finally:
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        _lib.check(lib.ctts_k_attention_cfg(0, ncu, 4), "attention_cfg")     # back to the shipped default: one workgroup per unit
        _lib.check(lib.ctts_k_attention_heads_per_wg(1), "heads_per_wg")


@pytest.mark.parametrize("n_live", [1, 5, 16, 17, 45, 64])
def test_attention_oproj_fused(G, n_live):
    """decode attention of the perf mode with o_proj + residual folded into the launch (attention_k<OPJ>, ctts_gpt_weights.wo_hd): every
    (utterance, head) unit multiplies its bf16-rounded output by its 64 columns of Wo, the row's last arriver adds the 12 partials onto
    the residual in head order and writes x (f32), its bf16 copy in fragment order and the 48 partial sums of squares -- against a
    float64 reference built from the unfused kernel's own bf16 output; rows beyond the live count / finished rows stay untouched;
    three launches in a row re-use the partial buffer and the counters; the result does not depend on the arrival order (two runs of
    the same launch are bit-identical)."""
    from chattts_amd.engine import pack_wo_heads, unpack_frag
    lib = _lib.lib()
    rs = np.random.RandomState(300 + n_live)
    B, nh, d, H, cmax = 64, 12, 64, 768, 640
    Bp = 64
    kc = G.dev(G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32)), torch.bfloat16)
    vc = G.dev(G.bf16_round(rs.standard_normal((B, nh, cmax, d)).astype(f32)), torch.bfloat16)
    wo = G.bf16_round((rs.standard_normal((H, H)) * 0.05).astype(f32))
    wo_hd = pack_wo_heads(torch.from_numpy(wo).to(torch.bfloat16)).to(G.DEV)
    part = torch.full((Bp * nh * H,), float("nan"), dtype=torch.float32, device=G.DEV)
    cnt = torch.zeros((Bp,), dtype=torch.int32, device=G.DEV)
    na = G.dev(np.array([n_live], np.int32))
    for rep in range(3):
        slots_b = rs.permutation(B)[:n_live]
        jlo = rs.randint(0, 30, size=n_live)
        slot = np.array([rs.randint(jlo[m] + 1, cmax) if rs.rand() < 0.8 else jlo[m] + rs.randint(0, 12) for m in range(n_live)])
        desc = np.zeros((Bp, 4), np.int32)
        desc[:, 0] = -1
        desc[:n_live, 0], desc[:n_live, 1], desc[:n_live, 2], desc[:n_live, 3] = slots_b, slot, slot - jlo, jlo
        dead = -1
        if n_live > 4 and rep == 1:          # a row that finished since the last compaction: its descriptor says -1, nothing is written
            dead = 2
            desc[dead, 0] = -1
        qkv = rs.standard_normal((Bp, 3 * H)).astype(f32)
        x0 = rs.standard_normal((Bp, H)).astype(f32)
        q_d, desc_d = G.dev(qkv), G.dev(desc)
        # the unfused kernel's bf16 attention output = the operand the o_proj sees
        o = torch.full((Bp * H,), float("nan"), dtype=torch.bfloat16, device=G.DEV)
        _lib.check(lib.ctts_k_attention_dec(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), cmax, o.data_ptr(), desc_d.data_ptr(),
                                            na.data_ptr(), Bp, None, None, 0, None), "attention_dec")
        att = unpack_frag(o.float().cpu(), Bp, H).numpy().astype(np.float64)
        res = []
        for _ in range(2):
            x = G.dev(x0.copy())
            xp = torch.full((Bp * H,), float("nan"), dtype=torch.bfloat16, device=G.DEV)
            ssq = torch.full((Bp, 48), float("nan"), dtype=torch.float32, device=G.DEV)
            _lib.check(lib.ctts_k_attention_oproj(q_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), cmax, wo_hd.data_ptr(), desc_d.data_ptr(),
                                                  na.data_ptr(), Bp, part.data_ptr(), cnt.data_ptr(), x.data_ptr(), xp.data_ptr(),
                                                  ssq.data_ptr(), None), "attention_oproj")
            torch.cuda.synchronize()
            res.append((x.cpu().numpy(), unpack_frag(xp.float().cpu(), Bp, H).numpy(), ssq.cpu().numpy()))
        assert int(cnt.abs().sum()) == 0
        (xa, xba, sa), (xb_, xbb, sb) = res
        live = np.array([m for m in range(n_live) if m != dead], np.int64)
        gone = np.array([m for m in range(Bp) if m >= n_live or m == dead], np.int64)
        assert np.array_equal(xa[live], xb_[live]) and np.array_equal(sa[live], sb[live])      # arrival order does not matter
        want = x0.astype(np.float64) + att @ wo.astype(np.float64).T
        assert np.abs(xa[live] - want[live]).max() < 2e-4, np.abs(xa[live] - want[live]).max()
        assert np.array_equal(xba[live], G.bf16_round(xa[live]))                                   # the bf16 copy is the rounded f32 row
        assert np.abs(sa[live] - (xa[live].astype(np.float64) ** 2).reshape(len(live), 48, 16).sum(-1)).max() < 1e-3
        assert np.array_equal(xa[gone], x0[gone])                                                  # absent rows: untouched
        assert np.isnan(sa[gone]).all()



# ------------------------------------------------------------------------------------------------
def test_embed_and_final_norm(G):
    lib = _lib.lib()
    rs = np.random.RandomState(3)
    B, tcap, T, max_new = 5, 20, 6, 14
    emb = rs.standard_normal((4, 626, 768)).astype(f32)
    ids = rs.randint(0, 626, size=(B, tcap, 4)).astype(np.int64)
    lens = np.array([7, 8, 9, 12, 19], np.int32)
    x = torch.empty((B, 768), dtype=torch.float32, device=G.DEV)
    e_d, i_d, l_d = G.dev(emb), G.dev(ids), G.dev(lens)
    _lib.check(lib.ctts_k_embed_codes(e_d.data_ptr(), i_d.data_ptr(), tcap, l_d.data_ptr(), x.data_ptr(), B, None), "embed")
    ref = np.stack([((emb[0][ids[b, lens[b] - 1, 0]] + emb[1][ids[b, lens[b] - 1, 1]]) + emb[2][ids[b, lens[b] - 1, 2]])
                    + emb[3][ids[b, lens[b] - 1, 3]] for b in range(B)])
    assert np.array_equal(x.cpu().numpy(), ref)  # same k-ordered f32 adds -> bit exact
    # final norm, decode layout (q_per_b = 1) and prefill layout (q_per_b = T)
    w = (1 + 0.1 * rs.standard_normal(768)).astype(f32)
    for qpb in (1, T):
        xin = rs.standard_normal((B * qpb, 768)).astype(f32) * 3
        hfin = torch.empty((B, 768), dtype=torch.float32, device=G.DEV)
        hid = torch.zeros((B, max_new, 768), dtype=torch.float32, device=G.DEV)
        x_d, w_d = G.dev(xin), G.dev(w)
        _lib.check(lib.ctts_k_final_norm(x_d.data_ptr(), qpb, w_d.data_ptr(), 1e-6, hfin.data_ptr(), hid.data_ptr(), max_new,
                                         l_d.data_ptr(), T, B, None), "final_norm")
        last = xin.reshape(B, qpb, 768)[:, -1]
        refn = llama_np.rmsnorm(last, w, f32(1e-6))
        assert np.abs(hfin.cpu().numpy() - refn).max() < 2e-6
        h = hid.cpu().numpy()
        for b in range(B):
            g = lens[b] - T
            assert np.array_equal(h[b, g], hfin.cpu().numpy()[b])


# ------------------------------------------------------------------------------------------------
# fused sampling kernel vs the reference's own outputs (goldens) and the oracle
# ------------------------------------------------------------------------------------------------
def run_sample_kernel(G, logits, hist, temp4, q, *, top_p, top_k, rep, mask_eos, row_offset=0, stop_at=None, margin=False, row_base=None):
    lib = _lib.lib()
    rows, V = logits.shape
    B = rows // 4
    h = hist.shape[1]
    T = 1
    tcap = T + h + 2
    ids = np.zeros((B, tcap, 4), np.int64)
    if h:
        ids[:, T: T + h, :] = hist.reshape(B, 4, h).transpose(0, 2, 1)
    keep = []
    d = lambda a: (keep.append(G.dev(a)), keep[-1])[1]
    s = _lib.GenState()
    s.B, s.T, s.max_new = B, T, h + 2
    ids_d = d(ids)
    len_d = d(np.full(B, T + h, np.int32))
    fin_d = d(np.zeros(B, np.uint8))
    end_d = d(np.zeros(B, np.int32))
    s.ids_buf, s.len, s.finish, s.end_idx = ids_d.data_ptr(), len_d.data_ptr(), fin_d.data_ptr(), end_d.data_ptr()
    s.q, s.nq = d(q.reshape(1, rows, V)).data_ptr(), 1
    s.temperature = d(temp4.astype(f32)).data_ptr()
    pt = rng.penalty_table(rep)
    s.pow_table = None if pt is None else d(pt.numpy()).data_ptr()
    s.top_p_thr = float(np.float32(1.0 - top_p)) if top_p is not None else 0.0
    s.use_top_p, s.top_k, s.use_top_k = int(top_p is not None), int(top_k or 0), int(top_k is not None)
    s.min_new = (h + 1) if mask_eos else 0
    s.eos, s.row_offset = 625, row_offset
    s.stop_at = None if stop_at is None else d(stop_at.astype(np.int32)).data_ptr()
    mg_d = d(np.full(B, np.inf, f32)) if margin else None
    s.margin = None if mg_d is None else mg_d.data_ptr()
    s.row_base = None if row_base is None else d(np.asarray(row_base, np.int32)).data_ptr()
    lg = d(logits.reshape(B, 4 * V))
    _lib.check(lib.ctts_k_sample(C.byref(s), lg.data_ptr(), None), "sample")
    torch.cuda.synchronize()
    out = ids_d.cpu().numpy()[:, T + h, :].reshape(-1)
    if margin:
        return out, mg_d.cpu().numpy()
    return out, fin_d.cpu().numpy(), end_d.cpu().numpy(), len_d.cpu().numpy()


@pytest.mark.parametrize("name", list(cases.SAMPLING_CASES))
def test_sample_vs_reference_golden(G, golden, name):
    c = cases.SAMPLING_CASES[name]
    logits, hist, temp = cases.sampling_inputs(c)
    rows, V = logits.shape
    q = rng.ExpDraws(rows, V, c["seed"]).step(0).numpy()
    got, fin, end, lens = run_sample_kernel(G, logits, hist, temp[:4], q, top_p=c["top_P"], top_k=c["top_K"], rep=c["rep"],
                                            mask_eos=c["mask_eos"])
    want = golden["sampling"][name + ".idx"]
    assert np.array_equal(got, want), (name, int((got != want).sum()))
    assert np.array_equal(fin, (want.reshape(-1, 4) == 625).any(1).astype(np.uint8))
    assert np.array_equal(end, 1 - fin)
    assert (lens == 1 + hist.shape[1] + 1).all()


def test_sample_row_offset_and_stop(G):
    """sharded rows keep the global rows>=625 quirk; stop_at forces / masks EOS like the oracle hook"""
    c = cases.SAMPLING_CASES["rows640"]
    logits, hist, temp = cases.sampling_inputs(c)
    q = rng.ExpDraws(640, 626, c["seed"]).step(0).numpy()
    pt = rng.penalty_table(c["rep"]).numpy()
    sl = slice(600, 640)
    got, *_ = run_sample_kernel(G, logits[sl], hist[sl], temp[:4], q[sl], top_p=c["top_P"], top_k=c["top_K"], rep=c["rep"],
                                mask_eos=False, row_offset=600)
    want = sampling_np.sample_step(logits[sl], hist[sl], q[sl], temperature=temp[sl], top_p=c["top_P"], top_k=c["top_K"],
                                   pow_table=pt, max_input_ids=625, row_offset=600)
    assert np.array_equal(got, want)
    # stop_at: rows of batch element 0 forced (gen=12 >= 5), element 1 masked (gen < 50), element 2 free
    stop = np.array([5, 50, -1] + [-1] * 7, np.int32)
    sl = slice(0, 40)
    got, fin, end, _ = run_sample_kernel(G, logits[sl], hist[sl], temp[:4], q[sl], top_p=c["top_P"], top_k=c["top_K"], rep=c["rep"],
                                         mask_eos=False, stop_at=stop)
    sa = np.repeat(stop, 4)
    want = sampling_np.sample_step(logits[sl], hist[sl], q[sl], temperature=temp[sl], top_p=c["top_P"], top_k=c["top_K"],
                                   pow_table=pt, max_input_ids=625, mask_eos=(sa >= 0) & (12 < sa), force_eos=(sa >= 0) & (12 >= sa))
    assert np.array_equal(got, want)
    assert (got[:4] == 625).all() and fin[0] == 1 and end[0] == 0


@pytest.mark.parametrize("trial", range(24))
def test_sample_randomised_vs_oracle(G, trial):
    """seeded sweep over the sampling parameters (temperature per codebook, top-p, top-k incl. None / > V, penalty,
    history length, EOS masking, tied logits): fused kernel == numpy oracle of the reference chain"""
    rs = np.random.RandomState(1000 + trial)
    B = int(rs.choice([1, 3, 8]))
    rows = B * 4
    scale = float(rs.choice([0.3, 1.0, 4.0, 8.0]))
    logits = (rs.standard_normal((rows, 626)) * scale).astype(f32)
    if trial % 5 == 0:
        logits = (np.round(logits * 2) / 2).astype(f32)  # many exact ties
    h = int(rs.choice([0, 1, 7, 16, 23]))
    hist = rs.randint(0, 626, size=(rows, h)).astype(np.int64)
    if h:
        hist[:, : h // 2] = np.argsort(-logits, axis=1)[:, : h // 2]  # penalise likely tokens too
    temp4 = rs.choice([0.1, 0.3, 0.7, 1.0, 1.5], size=4).astype(f32)
    top_p = [None, 0.05, 0.5, 0.7, 0.95, 0.999][int(rs.randint(6))]
    top_k = [None, 1, 3, 20, 100, 1000][int(rs.randint(6))]
    rep = [None, 1.05, 1.3, 2.0][int(rs.randint(4))]
    if trial % 5 == 0 and top_p is not None:
        top_p = None   # tie order inside the top-p cut is unspecified in the reference (unstable sort); ties are tested with top-k only
    mask_eos = bool(rs.randint(2))
    q = rng.ExpDraws(rows, 626, int(rs.randint(1 << 30))).step(0).numpy()
    got, *_ = run_sample_kernel(G, logits, hist, temp4, q, top_p=top_p, top_k=top_k, rep=rep, mask_eos=mask_eos)
    pt = rng.penalty_table(rep)
    want = sampling_np.sample_step(logits, hist, q, temperature=np.tile(temp4, B), top_p=top_p, top_k=top_k,
                                   pow_table=None if pt is None else pt.numpy(), max_input_ids=625, mask_eos=mask_eos)
    assert np.array_equal(got, want), (trial, top_p, top_k, rep, h, int((got != want).sum()))


def _cert_trial(trial):
    rs = np.random.RandomState(7000 + trial)
    B = int(rs.choice([2, 5, 8]))
    rows = B * 4
    scale = float(rs.choice([0.1, 0.5, 2.0, 4.0]))
    logits = (rs.standard_normal((rows, 626)) * scale).astype(f32)
    h = int(rs.choice([0, 3, 16, 21]))
    hist = rs.randint(0, 626, size=(rows, h)).astype(np.int64)
    if h:
        hist[:, : h // 2] = np.argsort(-logits, axis=1)[:, : h // 2]
    temp4 = rs.choice([0.3, 0.7, 1.0], size=4).astype(f32)
    top_p = [None, 0.5, 0.7, 0.95][int(rs.randint(4))]
    top_k = [None, 3, 20, 100][int(rs.randint(4))]
    rep = [None, 1.05, 1.3][int(rs.randint(3))]
    mask_eos = bool(rs.randint(2))
    q = rng.ExpDraws(rows, 626, int(rs.randint(1 << 30))).step(0).numpy()
    return rs, B, logits, hist, temp4, top_p, top_k, rep, mask_eos, q


@pytest.mark.parametrize("trial", range(16))
def test_sample_certificate_matches_its_float64_restatement(G, trial):
    """ctts_gen_state.margin (the parity certificate of round 6): the kernel's per-utterance minimum over its 4 sampling rows == the
    float64 restatement of the definition (oracle/sampling_np.decision_margin), fast path (top-k <= 64) and serial path alike"""
    rs, B, logits, hist, temp4, top_p, top_k, rep, mask_eos, q = _cert_trial(trial)
    got_ids, got = run_sample_kernel(G, logits, hist, temp4, q, top_p=top_p, top_k=top_k, rep=rep, mask_eos=mask_eos, margin=True)
    pt = rng.penalty_table(rep)
    want = sampling_np.decision_margin(logits, hist, q, temperature=np.tile(temp4, B), top_p=top_p, top_k=top_k,
                                       pow_table=None if pt is None else pt.numpy(), max_input_ids=625, mask_eos=mask_eos).reshape(B, 4).min(1)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), (got, want)
    assert np.allclose(got[fin], want[fin], rtol=2e-3, atol=2e-5), (trial, top_p, top_k, rep, got, want)


@pytest.mark.parametrize("trial", range(16))
def test_sample_certificate_is_a_certificate(G, trial):
    """what the margin promises: ANY perturbation of the logits that moves every tempered logit by less than margin / 2 leaves all
    4 sampled tokens of the utterance unchanged -- random-sign perturbations at 0.45 margin, 8 draws per trial"""
    rs, B, logits, hist, temp4, top_p, top_k, rep, mask_eos, q = _cert_trial(trial)
    kw = dict(top_p=top_p, top_k=top_k, rep=rep, mask_eos=mask_eos)
    base, mg = run_sample_kernel(G, logits, hist, temp4, q, margin=True, **kw)
    ok = np.isfinite(mg) & (mg > 1e-4)          # (below that, float32 rounding of the perturbed logits itself is not negligible)
    assert ok.any()
    eps = np.where(ok, 0.45 * mg, 0.0)          # tempered-logit units, per utterance
    for _ in range(8):
        sign = rs.choice([-1.0, 1.0], size=logits.shape) * rs.uniform(0.0, 1.0, size=logits.shape)
        delta = sign * np.repeat(eps, 4)[:, None] * np.tile(temp4, B)[:, None]     # raw-logit units of each row's temperature
        got, _ = run_sample_kernel(G, (logits + delta).astype(f32), hist, temp4, q, margin=True, **kw)
        same = (got.reshape(B, 4) == base.reshape(B, 4)).all(1)
        assert same[ok].all(), (trial, np.nonzero(~same & ok)[0], mg)


@pytest.mark.parametrize("trial", range(12))
def test_sample_text_kernel_vs_oracle_and_certificate(G, trial):
    """round 6: the rebuilt refine-text sampler alone (sample_text_k: 1024 threads, row in registers, column-maxima threshold, counting
    rank; serial extraction when there is no top-k, top-k > 64 or too many candidates) on random 21178-wide rows: the token == the numpy
    oracle of the reference chain (gpt.py:487-508 with one sampling row per utterance), the certificate == its float64 restatement, EOS
    masking, a row with many exact ties at the top."""
    lib = _lib.lib()
    rs = np.random.RandomState(9000 + trial)
    V = 21178
    B = int(rs.choice([1, 3, 6]))
    scale = float(rs.choice([0.5, 2.0, 4.0]))
    logits = (rs.standard_normal((B, V)) * scale).astype(f32)
    if trial % 4 == 3:
        logits = (np.round(logits * 4) / 4).astype(f32)          # exact ties everywhere, also at the k-th value
    temp = float(rs.choice([0.3, 0.7, 1.0]))
    top_p = [None, 0.5, 0.7, 0.95][int(rs.randint(4))]
    top_k = [20, 20, 3, 64, 100, None][int(rs.randint(6))]
    if trial % 4 == 3:
        top_p = None       # (tie order inside the top-p cut is unspecified in the reference: ties are tested with top-k only)
    if top_p is None and top_k is None:
        top_k = 20
    mask_eos = bool(rs.randint(2))
    eos = int(np.argmax(logits[0])) if mask_eos else 21000      # mask the likeliest token of row 0
    q = rng.ExpDraws(B, V, int(rs.randint(1 << 30))).step(0).numpy()
    T, tcap = 1, 4
    keep = []
    d = lambda a: (keep.append(G.dev(a)), keep[-1])[1]
    s = _lib.GenState()
    s.B, s.T, s.max_new = B, T, 3
    ids_d = d(np.zeros((B, tcap, 4), np.int64))
    len_d, fin_d, end_d = d(np.full(B, T, np.int32)), d(np.zeros(B, np.uint8)), d(np.zeros(B, np.int32))
    mg_d = d(np.full(B, np.inf, f32))
    s.ids_buf, s.len, s.finish, s.end_idx = ids_d.data_ptr(), len_d.data_ptr(), fin_d.data_ptr(), end_d.data_ptr()
    s.q, s.nq = d(q.reshape(1, B, V)).data_ptr(), 1
    s.temperature = d(np.array([temp], f32)).data_ptr()
    s.top_p_thr = float(np.float32(1.0 - top_p)) if top_p is not None else 0.0
    s.use_top_p, s.top_k, s.use_top_k = int(top_p is not None), int(top_k or 0), int(top_k is not None)
    s.min_new, s.eos, s.infer_text = (1 if mask_eos else 0), eos, 1
    s.margin = mg_d.data_ptr()
    _lib.check(lib.ctts_k_sample_text(C.byref(s), d(logits).data_ptr(), V, None), "sample_text")
    torch.cuda.synchronize()
    got = ids_d.cpu().numpy()[:, T, :]
    assert (got == got[:, :1]).all()                             # gpt.py:522-525: replicated over the 4 slots
    kw = dict(temperature=np.full(B, temp, f32), top_p=top_p, top_k=top_k, pow_table=None, max_input_ids=V - 1, mask_eos=mask_eos, eos=eos)
    want = sampling_np.sample_step(logits, np.zeros((B, 0), np.int64), q, **kw)
    assert np.array_equal(got[:, 0], want), (trial, top_p, top_k, temp, got[:, 0], want)
    assert np.array_equal(fin_d.cpu().numpy(), (want == eos).astype(np.uint8)) and (len_d.cpu().numpy() == T + 1).all()
    mg = mg_d.cpu().numpy()
    wm = sampling_np.decision_margin(logits, np.zeros((B, 0), np.int64), q, **kw)
    fin = np.isfinite(wm)
    assert np.array_equal(np.isfinite(mg), fin) and np.allclose(mg[fin], wm[fin], rtol=2e-3, atol=2e-5), (trial, top_p, top_k, mg, wm)


def test_sample_row_base_replaces_row_offset(G):
    """ctts_gen_state.row_base: per-utterance global sampling row (non-contiguous shards) -- the rows >= 625 penalty quirk follows it"""
    c = cases.SAMPLING_CASES["rows640"]
    logits, hist, temp = cases.sampling_inputs(c)
    q = rng.ExpDraws(640, 626, c["seed"]).step(0).numpy()
    pick = np.array([3, 157, 20, 159, 100, 156])          # utterances on both sides of sampling row 625 (= utterance 156, codebook 1)
    rows = (pick[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
    got, *_ = run_sample_kernel(G, logits[rows], hist[rows], temp[:4], q[rows], top_p=c["top_P"], top_k=c["top_K"], rep=c["rep"],
                                mask_eos=False, row_base=pick * 4)
    full = golden_rows640 = sampling_np.sample_step(logits, hist, q, temperature=temp, top_p=c["top_P"], top_k=c["top_K"],
                                                    pow_table=rng.penalty_table(c["rep"]).numpy(), max_input_ids=625)
    assert np.array_equal(got, full[rows])


# ------------------------------------------------------------------------------------------------
# codec streaming kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dil", [1, 2])
def test_dwconv_ln(G, dil):
    lib = _lib.lib()
    rs = np.random.RandomState(dil)
    B, F, Cc = 2, 29, 512
    x = rs.standard_normal((B, F, Cc)).astype(f32)
    w = rs.standard_normal((Cc, 1, 7)).astype(f32) * 0.4
    b = rs.standard_normal(Cc).astype(f32) * 0.1
    lw = (1 + 0.1 * rs.standard_normal(Cc)).astype(f32)
    lb = rs.standard_normal(Cc).astype(f32) * 0.1
    y = torch.empty((B, F, Cc), dtype=torch.float32, device=G.DEV)
    keep = [G.dev(x), G.dev(np.ascontiguousarray(w[:, 0, :].T)), G.dev(b), G.dev(lw), G.dev(lb)]
    _lib.check(lib.ctts_k_dwconv_ln(*[k.data_ptr() for k in keep], 1e-6, dil, y.data_ptr(), B, F, None), "dwconv_ln")
    ref = codec_np.layer_norm(codec_np.dwconv1d_cl(x, w, b, pad=3 * dil, dil=dil), lw, lb, 1e-6)
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-5
    y2 = torch.empty((B * F, Cc), dtype=torch.float32, device=G.DEV)
    _lib.check(lib.ctts_k_layernorm(keep[0].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), 1e-6, y2.data_ptr(), B * F, None), "ln")
    assert np.abs(y2.cpu().numpy() - codec_np.layer_norm(x, lw, lb, 1e-6).reshape(B * F, Cc)).max() < 2e-5


@pytest.mark.parametrize("dil,B,F", [(1, 13, 1000), (2, 13, 1000), (2, 7, 1900), (1, 200, 70)])
def test_dwconv_ln_sliding_window(G, dil, B, F):
    """from 12288 frames the depthwise conv + LayerNorm runs as `dwconv_ln_run_k` (a wave walks 36 frames of one utterance and phase,
    7 rows in a register ring): ragged run ends, utterances shorter than one run, both dilations, vs the numpy oracle -- and bit
    for bit the one-wave-per-frame kernel's result (same taps in the same order)."""
    lib = _lib.lib()
    rs = np.random.RandomState(dil + F)
    Cc = 512
    assert B * F >= 12288
    x = rs.standard_normal((B, F, Cc)).astype(f32)
    w = rs.standard_normal((Cc, 1, 7)).astype(f32) * 0.4
    b = rs.standard_normal(Cc).astype(f32) * 0.1
    lw = (1 + 0.1 * rs.standard_normal(Cc)).astype(f32)
    lb = rs.standard_normal(Cc).astype(f32) * 0.1
    y = torch.empty((B, F, Cc), dtype=torch.float32, device=G.DEV)
    keep = [G.dev(x), G.dev(np.ascontiguousarray(w[:, 0, :].T)), G.dev(b), G.dev(lw), G.dev(lb)]
    _lib.check(lib.ctts_k_dwconv_ln(*[k.data_ptr() for k in keep], 1e-6, dil, y.data_ptr(), B, F, None), "dwconv_ln")
    got = y.cpu().numpy()
    ref = codec_np.layer_norm(codec_np.dwconv1d_cl(x[:3], w, b, pad=3 * dil, dil=dil), lw, lb, 1e-6)
    assert np.abs(got[:3] - ref).max() < 2e-5
    # the small-batch kernel on slices of the same input (below the threshold) must give the same bits
    for lo in (0, B - 3):
        ys = torch.empty((3, F, Cc), dtype=torch.float32, device=G.DEV)
        xs = G.dev(x[lo: lo + 3])
        _lib.check(lib.ctts_k_dwconv_ln(xs.data_ptr(), *[k.data_ptr() for k in keep[1:]], 1e-6, dil, ys.data_ptr(), 3, F, None), "dwconv_ln small")
        assert np.array_equal(ys.cpu().numpy(), got[lo: lo + 3])


@pytest.mark.parametrize("B,F", [(1, 2), (2, 9), (3, 40)])
def test_istft(G, B, F):
    import math
    lib = _lib.lib()
    rs = np.random.RandomState(F)
    head = rs.standard_normal((B, F, 1026)).astype(f32)
    head[..., :513] = head[..., :513] * 0.7 + 0.5
    head[0, 0, 3] = 9.0  # exercises the exp clip at 1e2
    window = torch.hann_window(1024).numpy()
    kk = np.arange(512, dtype=np.float64) * (2 * math.pi / 1024)
    tw = np.stack([np.cos(kk), np.sin(kk)], 1).astype(f32)
    frames = torch.empty((B, F, 1024), dtype=torch.float32, device=G.DEV)
    wav = torch.empty((B, 256 * (F - 1)), dtype=torch.float32, device=G.DEV)
    keep = [G.dev(head), G.dev(window), G.dev(tw)]
    _lib.check(lib.ctts_k_istft(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), frames.data_ptr(), wav.data_ptr(), B, F, None), "istft")
    mag = np.minimum(np.exp(head[..., :513]), f32(1e2))
    spec = mag * (np.cos(head[..., 513:]) + 1j * np.sin(head[..., 513:]))
    ref = codec_np.istft_center(spec, window, 1024, 256)
    got = wav.cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


# ------------------------------------------------------------------------------------------------
# opt-in device generator for the multinomial's Exp(1) draws (ctts_gen_state.rng_device)
# ------------------------------------------------------------------------------------------------
def _device_draws(seed, step, row0, rows, V=626):
    lib = _lib.lib()
    out = torch.empty((rows, V), dtype=torch.float32, device="cuda:0")
    _lib.check(lib.ctts_k_exp_draws(seed, step, row0, rows, V, out.data_ptr(), None), "exp_draws")
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64)


def test_device_generator_draws_are_exp1(G):
    """the sampling kernel's own generator (Philox4x32-10 -> u in (0,1] -> -log u): Kolmogorov-Smirnov against Exp(1) on 160k
    draws, first two moments, and streams that differ by seed / step / row are distinct and uncorrelated"""
    q = _device_draws(12345, 7, 0, 256)
    x = np.sort(q.reshape(-1))
    n = x.size
    assert x[0] > 0 and np.isfinite(x).all()
    cdf = 1.0 - np.exp(-x)
    D = max(np.abs(cdf - np.arange(1, n + 1) / n).max(), np.abs(cdf - np.arange(0, n) / n).max())
    assert D < 1.63 / np.sqrt(n), (D, 1.63 / np.sqrt(n))        # 1 % critical value of the KS statistic
    assert abs(x.mean() - 1.0) < 0.01 and abs(x.var() - 1.0) < 0.03
    for other in (_device_draws(12346, 7, 0, 256), _device_draws(12345, 8, 0, 256), _device_draws(12345, 7, 256, 256)):
        assert not np.array_equal(other, q)
        r = np.corrcoef(other.reshape(-1), q.reshape(-1))[0, 1]
        assert abs(r) < 0.01, r
    # prefix property in the row index (what keeps an N-way sharded batch equal to the unsharded one): rows 100..131 of a draw
    # that starts at row 0 are the draw that starts at row 100
    assert np.array_equal(_device_draws(12345, 7, 100, 32), q[100:132])
    # neighbouring tokens of one row (the 4 outputs of one Philox call) are uncorrelated too
    assert abs(np.corrcoef(q[:, 0::4].reshape(-1), q[:, 1::4][:, : q[:, 0::4].shape[1]].reshape(-1))[0, 1]) < 0.01


def test_device_generator_token_histogram(G):
    """chi-square of 102,400 tokens sampled with the device generator from ONE fixed distribution (top-8 of a fixed logits row,
    temperature 1) against the exact softmax probabilities of the kept set: argmax(p / q) with q ~ Exp(1) IS a multinomial draw"""
    lib = _lib.lib()
    rs = np.random.RandomState(4)
    row = (rs.standard_normal(626) * 2.0).astype(f32)
    B, steps, V = 256, 100, 626
    logits = np.tile(row, (B * 4, 1))
    keep = []
    d = lambda a: (keep.append(G.dev(a)), keep[-1])[1]
    T, tcap = 1, 1 + steps + 1
    s = _lib.GenState()
    s.B, s.T, s.max_new = B, T, steps + 1
    ids_d = d(np.zeros((B, tcap, 4), np.int64))
    len_d, fin_d, end_d = d(np.full(B, T, np.int32)), d(np.zeros(B, np.uint8)), d(np.zeros(B, np.int32))
    s.ids_buf, s.len, s.finish, s.end_idx = ids_d.data_ptr(), len_d.data_ptr(), fin_d.data_ptr(), end_d.data_ptr()
    s.q, s.nq = None, 0
    s.rng_device, s.rng_per_step, s.rng_seed = 1, 1, d(np.array([987654321], np.int64)).data_ptr()
    s.temperature = d(np.ones(4, f32)).data_ptr()
    s.pow_table = None
    s.top_p_thr, s.use_top_p, s.top_k, s.use_top_k = 0.0, 0, 8, 1
    s.min_new, s.eos, s.row_offset = steps + 5, 625, 0        # EOS masked throughout: nothing finishes
    lg = d(logits.reshape(B, 4 * V))
    for _ in range(steps):
        _lib.check(lib.ctts_k_sample(C.byref(s), lg.data_ptr(), None), "sample")
    torch.cuda.synchronize()
    toks = ids_d.cpu().numpy()[:, T: T + steps, :].reshape(-1)
    top = np.argsort(-row.astype(np.float64), kind="stable")[:8]
    top = top[top != 625][:8]
    assert set(np.unique(toks)) <= set(top.tolist())
    p = np.exp(row[top].astype(np.float64) - row[top].max())
    p /= p.sum()
    obs = np.array([(toks == t).sum() for t in top], np.float64)
    exp = p * toks.size
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    print("device generator chi2 (7 dof):", chi2, obs.astype(int).tolist())
    assert toks.size == B * 4 * steps and chi2 < 24.3, chi2        # 0.1 % critical value at 7 degrees of freedom
    # per-step freshness: consecutive steps of one row are not the same token stream shifted / repeated
    t0 = ids_d.cpu().numpy()[:, T: T + steps, 0]
    assert (t0[:, 1:] != t0[:, :-1]).mean() > 0.5
