"""CPU oracle for the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain numpy (float32) restatement of the reference's algorithm for
    GPT.generate -> DVAE.forward(mode="decode") -> Vocos.decode
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; the product (`chattts_amd/`) never does and fails loudly without its HIP library.

Pinning status: the reference's own tests hold NO golden vectors for this path (SURVEY.md
section 4 / 8c: only a text-refine known-answer test that needs the real checkpoint).  The oracle is
therefore pinned against outputs of the reference code itself, run in the build container by
`oracle/make_goldens.py` (imports /root/reference, writes tests/golden/*.npz); see
tests/test_oracle_vs_golden.py.  Vocos is an un-vendored, un-installed dependency of the
reference: `codec_np.vocos_decode` restates its published algorithm (vocos.models.VocosBackbone,
vocos.heads.ISTFTHead, vocos.spectral_ops.ISTFT) and is pinned only against torch ops
(conv1d / layer_norm / istft) -- "parity unpinned against the vocos package" (DESIGN.md).
"""
