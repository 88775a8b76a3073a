"""Kernel-level parity: each gfx950 kernel vs a plain fp32/fp64 torch reference of the same op.

Tolerances (fp32 mode): the MFMA f32 path is an fmaf chain, so it differs from a float64
reference only by fp32 round-off: |err| <= 2e-5 * sqrt(K)-ish; we assert 1e-4 relative to the
output scale.  bf16 mode: operands rounded to 8 bits of mantissa -> 2e-2 relative.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


# fp32 contraction kernels: 0 = fp32 MFMA, 2 = exact 3-way bf16 split + 6 bf16 MFMA products
F32_MODES = [0, 2, 3]   # fp32 MFMA, bf16 three-term split, fp16 two-term split: one strict tolerance


def _rel_err(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-12)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 96), (257, 64, 64), (1000, 32, 288),
                                   (77, 153, 1536), (513, 1770, 1024), (64, 11, 256)])
@pytest.mark.parametrize("prec", F32_MODES)
def test_gemm_plain(built_lib, gpu, M, N, K, prec):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    # asymmetric, non-identity weights so a transposed C-write cannot pass
    W = torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N)[:, None]
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec)
    torch.cuda.synchronize()
    assert _rel_err(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize("prec", F32_MODES)
def test_gemm_epilogues(built_lib, gpu, prec):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 333, 160, 128
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    base = A.double() @ W.double().T + bias.double()
    # gelu
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec, act=1)
    assert _rel_err(out.cpu(), torch.nn.functional.gelu(base)) < 1e-5
    # swish * 0.5 + residual
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec, act=2, alpha=0.5, R=R.to(gpu))
    assert _rel_err(out.cpu(), R.double() + 0.5 * base * torch.sigmoid(base)) < 1e-5
    # residual + post relu
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec, R=R.to(gpu), post_relu=True)
    assert _rel_err(out.cpu(), torch.relu(base + R.double())) < 1e-5
    # weighted-sum accumulate
    WS = torch.zeros(M, N, device=gpu)
    ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec, WS=WS, ws_w=0.3, ws_init=True, ldws=N)
    ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=prec, WS=WS, ws_w=-1.2, ws_init=False, ldws=N)
    assert _rel_err(WS.cpu(), (0.3 - 1.2) * base) < 1e-5


@pytest.mark.parametrize("prec", F32_MODES)
def test_gemm_conv1d_overlapping_rows(built_lib, gpu, prec):
    """conv1d(k=3, s=2) over channels-last rows == contraction with lda = s*C, K = k*C."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(5)
    Bn, T, Ci, Co, k, s = 2, 101, 32, 48, 3, 2
    x = torch.randn(Bn, Ci, T, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) * 0.1
    ref = torch.nn.functional.conv1d(x.double(), w.double(), stride=s)  # [B, Co, To]
    To = ref.shape[-1]
    xcl = x.permute(0, 2, 1).contiguous()  # [B, T, Ci]
    wp = w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()  # k index = j*Ci + ci
    out = torch.empty(Bn, To, Co, device=gpu)
    ops.gemm(xcl.to(gpu).view(-1), wp.to(gpu), M=To, N=Co, K=k * Ci, lda=s * Ci, C_out=out.view(-1),
             ldc=Co, nz=Bn, zdiv=1, zs=dict(a_z0=T * Ci, c_z0=To * Co), precision=prec)
    torch.cuda.synchronize()
    assert _rel_err(out.cpu().permute(0, 2, 1), ref) < 1e-5


@pytest.mark.parametrize("prec", F32_MODES)
def test_gemm_grouped_posconv_two_level_k(built_lib, gpu, prec):
    """grouped conv1d(k=16, groups=4, pad) via (kc, ldk) addressing and z-batching."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(6)
    Bn, L, D, G, k = 2, 50, 128, 4, 16
    cg = D // G
    x = torch.randn(Bn, L, D, generator=g)
    w = torch.randn(D, cg, k, generator=g) * 0.1
    bias = torch.randn(D, generator=g)
    ref = torch.nn.functional.conv1d(x.double().permute(0, 2, 1), w.double(), bias.double(),
                                     padding=k // 2, groups=G)[..., :-1].permute(0, 2, 1)
    Lp = L + k
    xpad = torch.zeros(Bn, Lp, D)
    xpad[:, k // 2:k // 2 + L] = x
    wp = w.permute(0, 2, 1).reshape(D, k * cg).contiguous()  # [co][j*cg + ci]
    out = torch.empty(Bn, L, D, device=gpu)
    ops.gemm(xpad.to(gpu).view(-1), wp.to(gpu), M=L, N=cg, K=k * cg, lda=D, kc=cg, ldk=D,
             ldw=k * cg, bias=bias.to(gpu), C_out=out.view(-1), ldc=D, nz=Bn * G, zdiv=G,
             zs=dict(a_z0=Lp * D, a_z1=cg, w_z1=cg * k * cg, c_z0=L * D, c_z1=cg, b_z1=cg), precision=prec)
    torch.cuda.synchronize()
    assert _rel_err(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize("prec", F32_MODES)
def test_gemm_rowoff_tables(built_lib, gpu, prec):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(8)
    M, N, K = 200, 64, 64
    buf = torch.randn(4096, generator=g)
    aoff = (torch.randperm(900, generator=g)[:M] * 4).to(torch.int32)
    coff = (torch.randperm(M, generator=g) * N).to(torch.int32)
    W = torch.randn(N, K, generator=g)
    A = torch.stack([buf[o:o + K] for o in aoff.tolist()])
    ref = A.double() @ W.double().T
    out = torch.zeros(M * N, device=gpu)
    ops.gemm(buf.to(gpu), W.to(gpu), M=M, N=N, K=K, lda=0, a_rowoff=aoff.to(gpu),
             c_rowoff=coff.to(gpu), C_out=out, ldc=N, precision=prec)
    got = out.cpu().view(M, N)[(coff // N).long()]
    assert _rel_err(got, ref) < 1e-5


def test_gemm_split_is_fp32_grade(built_lib, gpu):
    """The 3-way split kernel against a float64 product, next to the fp32 MFMA kernel on the same
    data (wide dynamic range activations): its error must not exceed the fp32 MFMA kernel's (x1.5),
    measured relative to sum_k |a||w| (the scale fp32 rounding errors live on)."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 1024, 384, 2048
    A = torch.randn(M, K, generator=g) * torch.exp(2.0 * torch.randn(M, K, generator=g))
    W = torch.randn(N, K, generator=g) * 0.05
    ref = A.double() @ W.double().T
    scale = A.double().abs() @ W.double().abs().T
    errs = {}
    for prec in F32_MODES:
        out = ops.gemm(A.to(gpu), W.to(gpu), precision=prec).cpu().double()
        e = (out - ref).abs() / scale
        errs[prec] = (e.max().item(), e.pow(2).mean().sqrt().item())
    print("error vs float64 / sum|a||w|  (max, rms):", {{0: "fp32 MFMA", 2: "bf16x3 (6 products)", 3: "fp16x2 (3 products)"}[k]: v
                                                      for k, v in errs.items()})
    assert errs[2][0] <= 1.5 * errs[0][0] and errs[2][1] <= 1.5 * errs[0][1], errs
    assert errs[2][0] < 2e-6, errs
    # the two-term fp16 split ("3xFP16"): 22 significant bits per operand -> must also stay at the level of the
    # hardware fp32 MFMA on this wide-dynamic-range data (activations span 5 decades inside one tensor)
    assert errs[3][0] <= 1.5 * errs[0][0] and errs[3][1] <= 1.5 * errs[0][1], errs
    # the split itself is exact: hi + mid + lo == x bit for bit
    W3 = ops.split_weights(W.to(gpu)).cpu().view(torch.bfloat16).float()      # [N, K/32, 3, 32]
    pos = torch.tensor([8 * ((k & 15) >> 2) + (k & 3) + 4 * (k >> 4) for k in range(32)])
    rec = (W3[:, :, 0] + W3[:, :, 1] + W3[:, :, 2])[:, :, pos].reshape(N, K)
    assert torch.equal(rec, W)
    # two-term fp16 planes: (hi + lo) * col_scale reproduces W to 2^-22 (power-of-two row scales are exact)
    W2, cs = ops.split_weights_h2(W.to(gpu))
    W2 = W2.cpu().view(torch.float16).double()                                  # [N, K/32, 2, 32]
    rec2 = ((W2[:, :, 0] + W2[:, :, 1])[:, :, pos].reshape(N, K)) * cs.cpu().double()[:, None]
    # (elements more than 2^17 below their row maximum keep a subnormal lo term: absolute floor 2^-25 scaled)
    assert ((rec2 - W.double()).abs() <= 2.0 ** -22 * W.double().abs() + 2.0 ** -25 * cs.cpu().double()[:, None]).all()
    assert torch.equal(torch.log2(cs.cpu()).round(), torch.log2(cs.cpu()))       # exact powers of two
    assert (W2[:, :, 0].abs().amax(dim=(1, 2)) < 2.0 ** 15).all() and (W2[:, :, 0].abs().amax(dim=(1, 2)) >= 2.0 ** 14).all()


@pytest.mark.parametrize("M,N,K", [(1024, 384, 2048), (399 * 3, 1024, 1024), (777, 64, 256), (300, 96, 544)])
def test_gemm_f16_single_term(built_lib, gpu, M, N, K):
    """DZN_PREC_F16 (gemm_split.hip NP = 1): both operands rounded ONCE to fp16 after the exact per-unit /
    per-row power-of-two scaling, one MFMA product, fp32 accumulate.  Reference = the float64 product of the operands
    rounded the same way (must agree to fp32-accumulation level, so nothing else is lost), and the plain float64
    product (error at the fp16 rounding level 2^-11 per operand, relative to sum |a||w|).  Wide dynamic range between
    rows of A is harmless: the scale is per window (here: one unit), fp16 keeps 11 bits below it."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 3.0
    W = torch.randn(N, K, generator=g) * 0.05 * torch.exp(torch.randn(N, 1, generator=g))
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), R=R.to(gpu), precision=4).cpu().double()
    out3 = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), R=R.to(gpu), precision=3).cpu().double()
    # operands as the kernel rounds them
    sa = 2.0 ** (14 - torch.floor(torch.log2(A.abs().max())))
    Ah = (A.double() * sa).to(torch.float16).double() / sa
    sw = 2.0 ** (14 - torch.floor(torch.log2(W.abs().amax(dim=1, keepdim=True))))
    Wh = (W.double() * sw).to(torch.float16).double() / sw
    ref_h = Ah @ Wh.T + bias.double() + R.double()
    ref = A.double() @ W.double().T + bias.double() + R.double()
    scale = A.double().abs() @ W.double().abs().T + 1.0
    assert ((out - ref_h).abs() / scale).max().item() < 2e-6          # exactly the rounded operands, fp32 accumulate
    e = ((out - ref).abs() / scale)
    assert e.max().item() < 2.0 ** -10 and e.pow(2).mean().sqrt().item() < 2.0 ** -12
    assert ((out3 - ref).abs() / scale).max().item() < 2e-6           # the two-term mode on the same data, for scale


@pytest.mark.parametrize("M,N,K,cfg", [(300, 200, 512, "auto"), (257, 64, 96, "auto"), (515, 384, 1056, "128x128"), (515, 384, 1056, "128x64"),
                                       (1000, 1024, 1024, "128x128"), (129, 96, 32, "128x128"), (640, 640, 448, "auto")])
def test_gemm_mx_cross_terms(built_lib, gpu, M, N, K, cfg):
    """(r5) DZN_PREC_F16 as csrc/gemm_mx.hip computes it: fp16 hi*hi + the two cross terms in fp8 e4m3 on
    v_mfma_scale_f32_32x32x64_f8f6f4 (block scales 2^0 / 2^-11).  (a) Against testkit/mx_emulation.py — the same operands
    rounded the same way, exact accumulation: agreement at fp32-accumulation level pins the device's operand layout, the scale
    bytes, the fp8 conversion and the plane packing all at once (K % 64 == 32 exercises the half-filled last group, N % 64 != 0
    and M % 128 != 0 the tile edges, both tile shapes are forced).  (b) Against the plain float64 product: the error must sit
    near 2^-15 of sum |a||w| — an order of magnitude under the single-term mode on the same data, which is what brings the
    segmentation model inside SURVEY 8d's reduced bar."""
    import ctypes
    from diarizen_amd import _lib, ops
    from testkit.mx_emulation import mx_gemm, single_term_gemm
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 3.0 * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) * 0.05 * torch.exp(torch.randn(N, 1, generator=g))
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    lib = _lib.load()
    lib.dzn_op_set_gemm_mx_cfg(cfg.encode())
    try:
        out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), R=R.to(gpu), precision=4, mx=True).cpu().double()
        out1 = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), R=R.to(gpu), precision=4).cpu().double()
    finally:
        lib.dzn_op_set_gemm_mx_cfg(b"auto")
    tail = bias.double() + R.double()
    emu = mx_gemm(A, W) + tail
    ref = A.double() @ W.double().T + tail
    scale = A.double().abs() @ W.double().abs().T + 1.0
    e_emu = ((out - emu).abs() / scale).max().item()
    e = (out - ref).abs() / scale
    e1 = (out1 - ref).abs() / scale
    print(f"[mx {M}x{N}x{K} {cfg}] vs emulation {e_emu:.2e}; vs float64 max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}; "
          f"single term max {e1.max().item():.2e} rms {e1.pow(2).mean().sqrt().item():.2e}")
    assert e_emu < 2e-6                                   # the same rounded operands, fp32 accumulation
    assert e.max().item() < 2.0 ** -13 and e.pow(2).mean().sqrt().item() < 2.0 ** -15
    assert e.pow(2).mean().sqrt().item() * 8 < e1.pow(2).mean().sqrt().item()
    assert ((single_term_gemm(A, W) + tail - out1).abs() / scale).max().item() < 2e-6   # the single-term path is still there


def test_gemm_mx_layernorm_folded_and_row_stats(built_lib, gpu):
    """the MX contraction behind the descriptor's other features: a folded LayerNorm (raw rows in, statistics applied in the
    epilogue), GELU, per-row statistics of the output for the next folded norm, the |max| tracker of the output"""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 700, 320, 1024
    x = torch.randn(M, K, generator=g) * 2.0 + 0.7
    gamma = 1.0 + 0.1 * torch.randn(K, generator=g)
    beta = 0.1 * torch.randn(K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    b = torch.randn(N, generator=g)
    Wf = W * gamma
    bf = b + W @ beta
    csum = Wf.sum(1)
    stats = ops.row_stats(x.to(gpu), K, 1e-5)
    camax = torch.zeros(1, device=gpu)
    out, st = ops.gemm(x.to(gpu), Wf.to(gpu), bias=bf.to(gpu), ln_stats=stats, ln_colsum=csum.to(gpu), act=1, precision=4, mx=True,
                       want_row_stats=True, c_amax=camax)
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double())
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"[mx folded LN] max err {err:.2e}")
    assert err < 4e-3          # operands rounded at ~2^-15 of sum |x||w| (~60 here), amplified by rstd
    mu, var = out.cpu().double().mean(1), out.cpu().double().var(1, unbiased=False)
    assert (st.cpu()[:, 0].double() - mu).abs().max().item() < 1e-4
    assert (st.cpu()[:, 1].double() - (var + 1e-5).rsqrt()).abs().max().item() / (var + 1e-5).rsqrt().max().item() < 1e-4
    assert abs(camax.item() - out.abs().max().item()) < 1e-6


def test_gemm_bf16(built_lib, gpu):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 300, 200, 256
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    ref = A.bfloat16().double() @ W.bfloat16().double().T
    out = ops.gemm(A.to(gpu), None, N=N, K=K, ldw=K, W16=W.bfloat16().to(gpu), precision=1)
    torch.cuda.synchronize()
    assert _rel_err(out.cpu(), ref) < 1e-4  # vs the same bf16-rounded operands
    assert _rel_err(out.cpu(), A.double() @ W.double().T) < 2e-2


@pytest.mark.parametrize("C,Cpad", [(153, 160), (1024, 1024), (256, 256), (211, 224), (512, 512)])
def test_layernorm(built_lib, gpu, C, Cpad):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(C)
    x = torch.zeros(77, Cpad)
    x[:, :C] = torch.randn(77, C, generator=g) * 3 + 1
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = torch.nn.functional.layer_norm(x[:, :C].double(), (C,), gamma.double(), beta.double())
    out = ops.layernorm(x.to(gpu), gamma.to(gpu), beta.to(gpu), C_true=C)
    assert (out.cpu()[:, :C].double() - ref).abs().max() < 2e-5
    assert (out.cpu()[:, C:] == 0).all()
    outg = ops.layernorm(x.to(gpu), gamma.to(gpu), beta.to(gpu), C_true=C, gelu=True)
    assert (outg.cpu()[:, :C].double() - torch.nn.functional.gelu(ref)).abs().max() < 2e-5


def _attn_ref(q, k, v, bias=None):
    s = (q.double() * 0.125) @ k.double().transpose(-1, -2)
    if bias is not None:
        s = s + bias.double()
    return torch.softmax(s, -1) @ v.double()


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("L", [64, 99, 399])
def test_attention_nobias(built_lib, gpu, L, prec):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(L)
    B, h = 2, 4
    qkv = torch.randn(B * L, 3 * h * 64, generator=g)
    q, k, v = [qkv[:, i * h * 64:(i + 1) * h * 64].view(B, L, h, 64).permute(0, 2, 1, 3) for i in range(3)]
    ref = _attn_ref(q, k, v).permute(0, 2, 1, 3).reshape(B * L, h * 64)
    out = ops.attention(qkv.to(gpu), B, L, h, precision=prec)
    assert (out.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("L", [99, 399])
def test_attention_gated_relpos(built_lib, gpu, L, prec):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(L + 1)
    B, Htot = 2, 16
    heads = [1, 4, 7, 12, 13]
    h = len(heads)
    qkv = torch.randn(B * L, 3 * h * 64, generator=g)
    gate = torch.rand(B * L, Htot, generator=g) * 2
    table = torch.randn(Htot, 2 * L - 1, generator=g)
    q, k, v = [qkv[:, i * h * 64:(i + 1) * h * 64].view(B, L, h, 64).permute(0, 2, 1, 3) for i in range(3)]
    idx = torch.arange(L)[None, :] - torch.arange(L)[:, None] + L - 1  # key - query + L - 1
    P = table[:, idx]  # [Htot, L, L]
    bias = gate.view(B, L, Htot).permute(0, 2, 1)[..., None] * P[None]  # [B, Htot, L, L]
    bias = bias[:, heads]
    ref = _attn_ref(q, k, v, bias).permute(0, 2, 1, 3).reshape(B * L, h * 64)
    out = ops.attention(qkv.to(gpu), B, L, h, gate=gate.to(gpu), table=table.to(gpu),
                        head_idx=torch.tensor(heads, dtype=torch.int32, device=gpu), Htot=Htot, precision=prec)
    assert (out.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("L", [15, 64, 99, 399])
def test_attention_query_less_wavefronts_skip_with_the_same_bits(built_lib, gpu, L, prec):
    """(r5) L = 399 = 6 x 64 + 15: wavefronts of the last query tile that own no query skip scores / softmax / P.V
    (csrc/attention_split.hip).  Stored bits equal the kernel that computes them (dzn_op_set_attention_noskip), with and without
    the gated relative-position bias, for lengths with a partial last tile of every shape (15, 99, 399) and without one (64)."""
    from diarizen_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 * L + 1)
    B, Htot = 3, 16
    heads = [0, 3, 7, 12, 15]
    h = len(heads)
    qkv = (torch.randn(B * L, 3 * h * 64, generator=g) * 1.7).to(gpu)
    gate = (torch.rand(B * L, Htot, generator=g) * 2).to(gpu)
    table = torch.randn(Htot, 2 * L - 1, generator=g).to(gpu)
    hidx = torch.tensor(heads, dtype=torch.int32, device=gpu)
    for kw in ({}, dict(gate=gate, table=table, head_idx=hidx, Htot=Htot)):
        try:
            lib.dzn_op_set_attention_noskip(1)
            full = ops.attention(qkv, B, L, h, precision=prec, **kw).clone()
        finally:
            lib.dzn_op_set_attention_noskip(0)
        skip = ops.attention(qkv, B, L, h, precision=prec, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(skip).all()
        assert torch.equal(full, skip)


@pytest.mark.parametrize("L", [15, 64, 99, 399])
def test_attention_planes_kernel_matches_float64_and_the_split_kernel(built_lib, gpu, L):
    """(r6) csrc/attention_planes.hip: K / V arrive as fp16 two-term planes with ONE power-of-two scale per (row, head) - what
    the q/k/v contraction's epilogue writes - the scores are un-scaled per key and V's scale rides in the probabilities.  Against
    a float64 attention (W2V/components.py:453-486 with the gated relative-position bias of :690-725) at the tolerance of the
    in-kernel-split kernel, and against that kernel itself.  Rows of very different magnitude (x 2^-9 ... 2^9 per row and head:
    what per-row scales exist for) are part of the input; a transposed or permuted V gather would show as O(1)."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(3 * L + 2)
    B, Htot = 3, 16
    heads = [1, 4, 7, 12, 13]
    h = len(heads)
    qkv = torch.randn(B * L, 3 * h * 64, generator=g)
    # per-(row, head) magnitudes over 18 binades for K and V (the q side keeps one window scale)
    mag = torch.exp2(torch.randint(-9, 10, (B * L, 2 * h), generator=g).float())
    qkv[:, h * 64:] = (qkv[:, h * 64:].view(B * L, 2 * h, 64) * mag[..., None]).view(B * L, 2 * h * 64)
    qkv[:, :h * 64] *= 0.05                              # keep the softmax from saturating on the large keys
    gate = torch.rand(B * L, Htot, generator=g) * 2
    table = torch.randn(Htot, 2 * L - 1, generator=g)
    q, k, v = [qkv[:, i * h * 64:(i + 1) * h * 64].view(B, L, h, 64).permute(0, 2, 1, 3) for i in range(3)]
    idx = torch.arange(L)[None, :] - torch.arange(L)[:, None] + L - 1
    bias = (gate.view(B, L, Htot).permute(0, 2, 1)[..., None] * table[:, idx][None])[:, heads]
    hidx = torch.tensor(heads, dtype=torch.int32, device=gpu)
    for kw, b_ in ((dict(), None), (dict(gate=gate.to(gpu), table=table.to(gpu), head_idx=hidx, Htot=Htot), bias)):
        ref = _attn_ref(q, k, v, b_).permute(0, 2, 1, 3).reshape(B * L, h * 64)
        old = ops.attention(qkv.to(gpu), B, L, h, precision=_lib_prec("f32h"), **kw)
        outs = {}
        for qb, pf in ((1, 1), (1, 0), (2, 0)):  # the shipped form (64 queries per workgroup, next tile prefetched), without the
            try:                                 # prefetch, and 128 queries per workgroup
                _lib_handle().dzn_op_set_attention_qb(qb)
                _lib_handle().dzn_op_set_attention_prefetch(pf)
                outs[(qb, pf)] = ops.attention_planes(qkv.to(gpu), B, L, h, **kw).clone()
            finally:
                _lib_handle().dzn_op_set_attention_qb(1)
                _lib_handle().dzn_op_set_attention_prefetch(1)
        torch.cuda.synchronize()
        tol = 2e-5 * max(1.0, float(ref.abs().max()))
        e_old = (old.cpu().double() - ref).abs().max().item()
        for qb, out in outs.items():
            assert torch.isfinite(out).all()
            e_new = (out.cpu().double() - ref).abs().max().item()
            print(f"[attention planes L={L} bias={b_ is not None} qb={qb}] max err vs float64: planes {e_new:.2e}, in-kernel split "
                  f"{e_old:.2e} (|out| max {float(ref.abs().max()):.1f})")
            assert e_new < tol, (qb, e_new, tol)
        assert torch.equal(outs[(1, 1)], outs[(1, 0)]) and torch.equal(outs[(1, 1)], outs[(2, 0)])    # schedules, not arithmetic


def _lib_handle():
    from diarizen_amd import _lib
    return _lib.load()


def _lib_prec(name):
    from diarizen_amd import _lib
    return {"f32h": _lib.DZN_PREC_F32_H2, "f32s": _lib.DZN_PREC_F32_SPLIT, "f32": _lib.DZN_PREC_F32}[name]


@pytest.mark.parametrize("M,N,K,col0", [(399 * 2, 3 * 5 * 64, 256, 5 * 64), (1000, 3 * 16 * 64, 1024, 16 * 64), (130, 192, 64, 64)])
def test_gemm_epilogue_writes_kv_planes(built_lib, gpu, M, N, K, col0):
    """(r6) dzn_gemm_desc.kv_planes: the f32h contraction stores the columns >= kv_col0 as fp16 two-term planes with one exact
    power-of-two scale per (row, 64-column slot) and leaves the columns below as fp32.  (hi + lo) * inv reproduces the plain
    launch's fp32 value to 2^-21 of the slot's |max| (two fp16 terms = 22 bits), the slot |max| lands in [2^14, 2^15) and the Q
    columns are bit-identical to the plain launch; with a bias and a folded LayerNorm as the engine's q/k/v site has them."""
    from diarizen_amd import _lib, ops
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g).float())).to(gpu)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    prec = _lib.DZN_PREC_F32_H2
    plain = ops.gemm(A, W, bias=bias, precision=prec)
    C, planes, inv = ops.gemm(A, W, bias=bias, precision=prec, kv_col0=col0)
    torch.cuda.synchronize()
    assert torch.equal(C[:, :col0], plain[:, :col0])
    hi = planes[0].view(torch.float16).float()
    lo = planes[1].view(torch.float16).float()
    S = (N - col0) // 64
    rec = ((hi + lo).view(M, S, 64) * inv[..., None]).view(M, N - col0)
    want = plain[:, col0:]
    slot_max = want.view(M, S, 64).abs().amax(-1, keepdim=True)
    err = ((rec - want).view(M, S, 64).abs() / slot_max.clamp_min(1e-30)).max().item()
    scaled = (hi.view(M, S, 64).abs().amax(-1))
    assert err <= 2.0 ** -21, err
    assert float(scaled.min()) >= 2.0 ** 14 - 8 and float(scaled.max()) < 2.0 ** 15 + 1
    assert torch.all(inv > 0) and torch.equal(inv, torch.exp2(torch.floor(torch.log2(inv))))     # exact powers of two


def test_gate(built_lib, gpu):
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, Htot = 101, 16
    y = torch.randn(rows, Htot * 64, generator=g)
    Wg, bg = torch.randn(8, 64, generator=g) * 0.2, torch.randn(8, generator=g)
    cst = torch.rand(Htot, generator=g) + 0.5
    t = (y.double().view(rows, Htot, 64) @ Wg.double().T + bg.double()).view(rows, Htot, 2, 4).sum(-1)
    a, b = torch.sigmoid(t)[..., 0], torch.sigmoid(t)[..., 1]
    ref = a * (b * cst.double() - 1.0) + 2.0
    out = ops.gate(y.to(gpu), Wg.to(gpu), bg.to(gpu), cst.to(gpu))
    assert (out.cpu().double() - ref).abs().max() < 1e-5


def test_gate_stats_fused_layernorm(built_lib, gpu):
    """pre-norm fusion: one pass over the raw residual rows gives the LayerNorm statistics AND the gate evaluated
    on LayerNorm(x) (W2V/components.py:702-710 after :923), vs float64; rows with a large common offset too."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(12)
    rows, Htot = 203, 16
    x = torch.randn(rows, Htot * 64, generator=g) * 2.5 + torch.randn(rows, 1, generator=g) * 4
    gamma, beta = 1 + 0.2 * torch.randn(Htot * 64, generator=g), 0.2 * torch.randn(Htot * 64, generator=g)
    Wg, bg = torch.randn(8, 64, generator=g) * 0.2, torch.randn(8, generator=g)
    cst = torch.rand(Htot, generator=g) + 0.5
    y = torch.nn.functional.layer_norm(x.double(), (Htot * 64,), gamma.double(), beta.double())
    t = (y.view(rows, Htot, 64) @ Wg.double().T + bg.double()).view(rows, Htot, 2, 4).sum(-1)
    a, b = torch.sigmoid(t)[..., 0], torch.sigmoid(t)[..., 1]
    ref = a * (b * cst.double() - 1.0) + 2.0
    gate, stats = ops.gate_stats(x.to(gpu), gamma.to(gpu), beta.to(gpu), Wg.to(gpu), bg.to(gpu), cst.to(gpu))
    assert (gate.cpu().double() - ref).abs().max() < 2e-5
    mean = x.double().mean(1)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    assert (stats.cpu()[:, 0].double() - mean).abs().max() < 1e-5
    assert ((stats.cpu()[:, 1].double() - rstd) / rstd).abs().max() < 1e-5
    st2 = ops.row_stats(x.to(gpu))
    assert torch.equal(st2.cpu(), stats.cpu())


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("M,N,K,Kt", [(700, 960, 1024, 1024), (333, 1024, 256, 256), (500, 256, 224, 211)])
def test_gemm_layernorm_folded(built_lib, gpu, M, N, K, Kt, prec):
    """LayerNorm folded into the contraction (dzn_gemm_desc.ln_stats): raw rows in, W' = W diag(gamma),
    bias' = bias + W beta, epilogue rstd * (acc - mean * colsum(W')) -> == Linear(LayerNorm(x)) in float64,
    to fp32 grade, incl. rows with a mean 4x their spread and a padded K (feature projection: 211 -> 224)."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.zeros(M, K)
    x[:, :Kt] = torch.randn(M, Kt, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 6
    gamma, beta = 1 + 0.3 * torch.randn(Kt, generator=g), 0.3 * torch.randn(Kt, generator=g)
    W = torch.randn(N, Kt, generator=g) / Kt ** 0.5
    bias = torch.randn(N, generator=g)
    ref = torch.nn.functional.layer_norm(x[:, :Kt].double(), (Kt,), gamma.double(), beta.double()) @ W.double().T + bias.double()
    Wf = torch.zeros(N, K)
    Wf[:, :Kt] = W * gamma
    colsum = Wf.double().sum(1).float()
    bias_f = (bias.double() + W.double() @ beta.double()).float()
    stats = ops.row_stats(x.to(gpu), C_true=Kt)
    out = ops.gemm(x.to(gpu), Wf.to(gpu), bias=bias_f.to(gpu), precision=prec, ln_stats=stats, ln_colsum=colsum.to(gpu),
                   act=1)
    err = (out.cpu().double() - torch.nn.functional.gelu(ref)).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("M,N,K", [(700, 1024, 320), (333, 1024, 640), (515, 256, 1024)])
def test_gemm_epilogue_row_stats(built_lib, gpu, M, N, K, prec):
    """the contraction epilogue leaves the LayerNorm statistics (mean, rstd) of the rows it writes (partial sums per
    wavefront tile + stats_finalize_kernel): equal to a LayerNorm-statistics pass over the stored result"""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    R = torch.randn(M, N, generator=g) * 3 + torch.randn(M, 1, generator=g)
    out, stats = ops.gemm(A.to(gpu), W.to(gpu), R=R.to(gpu), precision=prec, want_row_stats=True)
    o = out.cpu().double()
    mean, rstd = o.mean(1), 1.0 / torch.sqrt(o.var(1, unbiased=False) + 1e-5)
    assert (stats.cpu()[:, 0].double() - mean).abs().max() < 2e-6 * (1 + mean.abs().max())
    assert ((stats.cpu()[:, 1].double() - rstd) / rstd).abs().max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (513, 1024, 1056), (257, 64, 96), (1000, 32, 288)])
def test_gemm_bf16_activations(built_lib, gpu, M, N, K):
    """both operands bf16 in HBM (LDS-DMA path), K tail at a 32-element half, bf16 / fp32 outputs,
    bf16 residual, GELU epilogue"""
    from conftest import needs_bf16_mode
    needs_bf16_mode(built_lib)      # gemm_lowp.hip: the quarantined bf16 engine mode
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).bfloat16()
    base = A.double() @ W.double().T + bias.double()
    out32 = torch.empty(M, N, device=gpu)
    ops.gemm(A.to(gpu), None, N=N, K=K, ldw=K, W16=W.to(gpu), bias=bias.to(gpu), C_out=out32, precision=1)
    assert _rel_err(out32.cpu(), base) < 1e-5          # exact bf16 products, fp32 accumulation
    out16 = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm(A.to(gpu), None, N=N, K=K, ldw=K, W16=W.to(gpu), bias=bias.to(gpu), C_out=out16, precision=1,
             act=1, R=R.to(gpu), post_relu=True)
    ref = torch.relu(torch.nn.functional.gelu(base) + R.double())
    assert _rel_err(out16.float().cpu(), ref) < 1e-2   # output rounded to bf16


def test_gemm_bf16_activations_conv_addressing(built_lib, gpu):
    """two-level K addressing + z batching on bf16 activations (positional-conv shape family)"""
    from conftest import needs_bf16_mode
    needs_bf16_mode(built_lib)
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(60)
    Bn, L, D, G, k = 2, 50, 256, 4, 16
    cg = D // G
    x = torch.randn(Bn, L, D, generator=g).bfloat16()
    w = (torch.randn(D, cg, k, generator=g) * 0.1).bfloat16()
    ref = torch.nn.functional.conv1d(x.double().permute(0, 2, 1), w.double(), None, padding=k // 2,
                                     groups=G)[..., :-1].permute(0, 2, 1)
    Lp = L + k
    xpad = torch.zeros(Bn, Lp, D, dtype=torch.bfloat16)
    xpad[:, k // 2:k // 2 + L] = x
    wp = w.permute(0, 2, 1).reshape(D, k * cg).contiguous()
    out = torch.empty(Bn, L, D, device=gpu)
    ops.gemm(xpad.to(gpu).view(-1), None, M=L, N=cg, K=k * cg, lda=D, kc=cg, ldk=D, ldw=k * cg,
             W16=wp.to(gpu), C_out=out.view(-1), ldc=D, nz=Bn * G, zdiv=G, precision=1,
             zs=dict(a_z0=Lp * D, a_z1=cg, w_z1=cg * k * cg, c_z0=L * D, c_z1=cg))
    torch.cuda.synchronize()
    assert _rel_err(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize("C", [40, 700, 2600])
def test_linkage_centroid_equals_scipy(built_lib, gpu, C):
    """csrc/linkage.hip vs scipy.cluster.hierarchy.linkage(method="centroid"): same dendrogram (ids, sizes
    exactly; distances to 1e-12) and the same flat clusters, on embeddings with speaker structure."""
    import numpy as np
    from scipy.cluster.hierarchy import fcluster, linkage
    from diarizen_amd import ops
    from oracle.gen_golden import synth_host_case
    seg, emb = synth_host_case(C, C=C, L=99, n_spk=5)
    e = emb[seg.sum(1) > 0].astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    Zs = linkage(e, method="centroid", metric="euclidean")
    Zg = ops.linkage_centroid(e)
    assert np.array_equal(Zs[:, [0, 1, 3]], Zg[:, [0, 1, 3]])
    assert np.abs(Zs[:, 2] - Zg[:, 2]).max() <= 1e-12
    for thr in (0.5, 0.7, 1.0):
        assert np.array_equal(fcluster(Zs, thr, "distance"), fcluster(Zg, thr, "distance"))


def test_linkage_centroid_30k_equals_scipy_golden(built_lib, gpu):
    """the device linkage at n >= 30 000 against scipy's dendrogram of the same embeddings
    (tests/golden/linkage_30k.npz, made by oracle/gen_golden.py linkage_scale — scipy needs minutes for it; the
    embeddings are regenerated here from elementwise float32 operations on seeded draws and checked by their md5)."""
    import hashlib
    from pathlib import Path
    import numpy as np
    from scipy.cluster.hierarchy import fcluster
    from diarizen_amd import ops
    from oracle.gen_golden import linkage_scale_case
    g = np.load(Path(__file__).parent / "golden" / "linkage_30k.npz")
    e = linkage_scale_case()
    assert len(e) >= 30000 and hashlib.md5(e.tobytes()).hexdigest() == str(g["emb_md5"])
    Zg = ops.linkage_centroid(e)
    assert np.array_equal(Zg[:, :2].astype(np.int32), g["ids"]) and np.array_equal(Zg[:, 3].astype(np.int32), g["size"])
    assert np.abs(Zg[:, 2] - g["dist"]).max() <= 1e-12 * g["dist"].max()
    Zs = np.column_stack([g["ids"].astype(np.float64), g["dist"], g["size"].astype(np.float64)])
    for thr in (0.8, 1.5, 2.5):
        assert np.array_equal(fcluster(Zs, thr, "distance"), fcluster(Zg, thr, "distance"))


def test_linkage_step_loop_equals_two_kernel_loop(built_lib, gpu, monkeypatch):
    """r3: one launch per step (merge or rescan, every workgroup selecting redundantly from the published records)
    against the r2 loop (single-workgroup selection + wide update): bit-identical dendrograms, also for n that is not a
    multiple of the 256-row block, n = 2, and duplicated embeddings (zero distances, ties broken by the lowest row)."""
    import numpy as np
    from diarizen_amd import ops
    from oracle.gen_golden import linkage_scale_case
    for n in (2, 3, 255, 256, 257, 1000, 4097):
        e = linkage_scale_case(n=n, dim=32, K=4, seed=n)
        if n >= 255:
            e[7] = e[3]
            e[100] = e[3]
        Za = ops.linkage_centroid(e)                 # r6 default: the step loop with two remembered neighbours per row
        monkeypatch.setenv("DZN_LINKAGE_PERSIST", "1")     # r6b: the same rules in ONE persistent launch (opt-in: measured slower)
        Zd = ops.linkage_centroid(e)
        monkeypatch.delenv("DZN_LINKAGE_PERSIST")
        assert np.array_equal(Za, Zd), n
        monkeypatch.setenv("DZN_LINKAGE_TWO_KERNEL", "1")
        Zb = ops.linkage_centroid(e)
        monkeypatch.delenv("DZN_LINKAGE_TWO_KERNEL")
        monkeypatch.setenv("DZN_LINKAGE_TOP1", "1")  # r3-r5: the step loop with one remembered neighbour
        Zc = ops.linkage_centroid(e)
        monkeypatch.delenv("DZN_LINKAGE_TOP1")
        assert np.array_equal(Zc, Zb), n
        assert np.array_equal(Za, Zb), n


def test_cdist_cosine_equals_scipy(built_lib, gpu):
    """csrc/linkage.hip dzn_cdist_cosine vs scipy.spatial.distance.cdist(metric="cosine") (the assignment step,
    PA/pipelines/clustering.py:207-216): float64 with in-order sums.  scipy's own summation order is the library
    build's (it is not the plain loop: an in-order numpy emulation differs from it by up to 2e-15 too), so the bar is
    |d| <= 1e-14 on distances of order 1, the same per-row argmin wherever the two best distances are more than
    1e-12 apart, the same NaN pattern (NaN row, zero row), and IDENTICAL scores for identical rows (the inactive
    speakers share one embedding; ties must stay ties)."""
    import json
    import os
    import numpy as np
    from scipy.spatial.distance import cdist
    from diarizen_amd import ops
    r = np.random.default_rng(11)
    n, dim, k = 20011, 256, 13
    cent = r.standard_normal((k, dim))
    lab = r.integers(0, k, n)
    e = (cent[lab] + 0.6 * r.standard_normal((n, dim))).astype(np.float32)
    e[5] = np.nan
    e[6] = 0.0
    e[100:140] = e[99]                                   # the inactive speakers share one embedding
    good = np.isfinite(e).all(axis=1)
    cent32 = np.vstack([e[(lab == j) & good].mean(axis=0) if j else e[7] for j in range(k)])   # float32; one == e[7]
    with np.errstate(invalid="ignore", divide="ignore"):
        want = cdist(e, cent32, metric="cosine")
    got = ops.cdist_cosine(e, cent32)
    assert np.array_equal(np.isnan(want), np.isnan(got)) and np.isnan(got[5]).all() and np.isnan(got[6]).all()
    ok = ~np.isnan(want)
    diff = np.abs(got[ok] - want[ok])
    exact = float(np.mean(got[ok] == want[ok]))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/cdist_vs_scipy.json", "w") as f:
            json.dump({"n": n, "k": k, "dim": dim, "bit_exact_fraction": exact, "max_abs_diff": float(diff.max())}, f)
    assert diff.max() <= 1e-14, (diff.max(), exact)
    rows = ok.all(axis=1)
    srt = np.sort(want[rows], axis=1)
    clear = (srt[:, 1] - srt[:, 0]) > 1e-12
    assert clear.mean() > 0.99
    assert np.array_equal(np.argmin(got[rows], axis=1)[clear], np.argmin(want[rows], axis=1)[clear])
    assert np.array_equal(got[100:140], np.broadcast_to(got[99], (40, k)))        # identical rows -> identical scores
    assert abs(got[7, 0]) < 1e-15 and abs(want[7, 0]) < 1e-15                     # a row that IS a centroid


@pytest.mark.parametrize("E,K", [(5000, 7), (20011, 40), (300, 3)])
def test_vbx_gmm_equals_numpy(built_lib, gpu, E, K):
    """csrc/vbx.hip (statistics + E-step on the device, float64) against the numpy loop of
    diarizen/clustering/VBx.py:99-107 as restated in clustering.vb_gmm: responsibilities to 1e-9, priors to 1e-10,
    identical hard decisions, and the same number of iterations (the ELBO stopping rule sees the same values)."""
    import numpy as np
    from diarizen_amd import clustering as cl
    from oracle.gen_golden import synth_vbx_case as _vbx_case
    X, Phi, q0 = _vbx_case(E, K)
    g_ref, pi_ref = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 20, backend="numpy")
    g_dev, pi_dev = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 20, backend="hip")
    assert g_dev.shape == g_ref.shape and np.isfinite(g_dev).all()
    assert np.abs(g_dev - g_ref).max() <= 1e-9
    assert np.abs(pi_dev - pi_ref).max() <= 1e-10
    assert np.array_equal(g_dev.argmax(1), g_ref.argmax(1))
    assert np.allclose(g_dev.sum(1), 1.0, atol=1e-12)
    # a single iteration (the stopping rule never fires): the E-step alone
    g1_ref, _ = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 1, backend="numpy")
    g1_dev, _ = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 1, backend="hip")
    assert np.abs(g1_dev - g1_ref).max() <= 1e-11


def test_vbx_states_share_the_host_arena_or_own_a_block(built_lib, gpu):
    """(r5) csrc/vbx.hip carves a state from the host stage's state arena (csrc/linkage.hip: no hipMalloc / hipFree per recording,
    own stream); a SECOND state created while the first lives gets a block of its own.  Both give the statistics of their own
    responsibilities (column sums of gamma, gamma^T rho) while the default stream is busy, the arena is reused by the next state
    and returned by dzn_host_workspace_release once no state holds it."""
    import ctypes as C
    import numpy as np
    from diarizen_amd import _lib
    from oracle.gen_golden import synth_vbx_case
    lib = _lib.load()

    def create(E, K):
        X, Phi, q0 = synth_vbx_case(E, K)
        X, Phi, q0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Phi, q0))
        st = C.c_void_p()
        _lib.check(lib.dzn_vbx_create(X.ctypes.data_as(C.c_void_p), Phi.ctypes.data_as(C.c_void_p), q0.ctypes.data_as(C.c_void_p),
                                      E, X.shape[1], K, 0, C.byref(st)), None, "dzn_vbx_create")
        return st, X, Phi, q0

    def stats(st, X, Phi, q0):
        K, D = q0.shape[1], X.shape[1]
        out = np.empty((K, D + 1))
        _lib.check(lib.dzn_vbx_stats(st, out.ctypes.data_as(C.c_void_p)), None, "dzn_vbx_stats")
        rho = X * np.sqrt(Phi)
        assert np.allclose(out[:, :D], q0.T @ rho, rtol=1e-12, atol=1e-10) and np.allclose(out[:, D], q0.sum(0), rtol=1e-12)

    lib.dzn_host_workspace_release(0)
    assert lib.dzn_host_workspace_bytes(0) == 0
    a = create(5000, 7)
    held = lib.dzn_host_workspace_bytes(0)
    assert held >= 5000 * 128 * 8 * 2
    b = create(3000, 5)                               # the arena is leased: a block of its own
    assert lib.dzn_host_workspace_bytes(0) == held
    big = torch.randn(8192, 8192, device=gpu)
    for _ in range(20):
        big = big @ big * 1e-4                        # default-stream work in flight while the states are used
    stats(*b)
    stats(*a)
    torch.cuda.synchronize()
    lib.dzn_host_workspace_release(0)                 # a live state keeps its arena
    assert lib.dzn_host_workspace_bytes(0) == held
    lib.dzn_vbx_destroy(a[0])
    lib.dzn_vbx_destroy(b[0])
    c = create(4000, 6)                               # reuses the arena (no growth)
    assert lib.dzn_host_workspace_bytes(0) == held
    stats(*c)
    lib.dzn_vbx_destroy(c[0])
    lib.dzn_host_workspace_release(0)
    assert lib.dzn_host_workspace_bytes(0) == 0


def test_clustering_backends_agree(built_lib, gpu):
    """AHC and VBx-style AHC initialisation through both linkage backends and both cdist backends: identical hard
    clusters."""
    import numpy as np
    from diarizen_amd import clustering as cl
    from oracle.gen_golden import synth_host_case
    seg, emb = synth_host_case(5, C=1500, L=99, n_spk=4)
    out = {}
    for backend in ("scipy", "hip"):
        ahc = cl.AgglomerativeClustering(threshold=0.7, min_cluster_size=13, linkage_backend=backend)
        ahc.cdist_backend = backend
        hard, soft, cent = ahc(embeddings=emb.astype(np.float32), segmentations=seg, min_clusters=1, max_clusters=20)
        out[backend] = (hard, cent, soft)
    assert np.array_equal(out["scipy"][0], out["hip"][0])
    assert np.allclose(out["scipy"][1], out["hip"][1])
    assert np.allclose(out["scipy"][2], out["hip"][2], rtol=0, atol=1e-14, equal_nan=True)


@pytest.mark.parametrize("H,W,B", [(5, 37, 2), (80, 798, 1), (12, 126, 3)])
def test_conv3x3_c32_split(built_lib, gpu, H, W, B):
    """csrc/conv_split.hip vs torch conv2d in float64: plain, and with ReLU / residual / post-ReLU; borders
    of the output image must stay exactly zero (they are the next layer's padding)."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    bias = torch.randn(32, generator=g)
    res = torch.randn(B, 32, H, W, generator=g)
    def padded(t):
        o = torch.zeros(B, H + 2, W + 2, 32)
        o[:, 1:-1, 1:-1] = t.permute(0, 2, 3, 1)
        return o.contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(32, 288).contiguous()           # k = (dh*3 + dw)*32 + ci
    W3 = ops.split_weights(wp.to(gpu))
    h2w = ops.split_weights_h2(wp.to(gpu))
    conv = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    cases = [(False, False, None, conv),
             (True, False, None, torch.relu(conv)),
             (False, True, res, torch.relu(conv + res.double()))]
    for relu, post, R, ref in cases:
        out = ops.conv3x3_c32(padded(x).to(gpu), W3, bias.to(gpu), R=None if R is None else padded(R).to(gpu),
                              relu=relu, post_relu=post).cpu()
        got = out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double()
        assert _rel_err(got, ref) < 1e-5
        assert out[:, 0].abs().max() == 0 and out[:, -1].abs().max() == 0
        assert out[:, :, 0].abs().max() == 0 and out[:, :, -1].abs().max() == 0
        # fp16 two-term variant of the same kernel (per-image power-of-two scales): same tolerance
        out2 = ops.conv3x3_c32(padded(x).to(gpu), W3, bias.to(gpu), R=None if R is None else padded(R).to(gpu),
                               relu=relu, post_relu=post, h2_weights=h2w).cpu()
        assert _rel_err(out2[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), ref) < 1e-5
        assert out2[:, 0].abs().max() == 0 and out2[:, :, 0].abs().max() == 0


@pytest.mark.parametrize("H,W,B", [(5, 37, 2), (80, 798, 2), (12, 126, 3), (3, 60, 1), (7, 61, 2), (1, 1, 1)])
def test_resblock32_fused_equals_two_convs(built_lib, gpu, H, W, B):
    """(r4) csrc/resblock_fused.hip — both convolutions of a 32-channel BasicBlock in one kernel, the intermediate image kept
    in LDS line buffers — against (a) the block evaluated by torch in float64 (wespeaker/resnet.py:139-144 with folded
    BatchNorm: relu(conv2(relu(conv1(x) + b1)) + b2 + x)) and (b) the same block as TWO launches of the per-conv kernel
    (conv_split.hip, fp16 two-term variant).  conv1 is bit-identical by construction; the only difference is the
    power-of-two scale of the intermediate's fp16 split (a-priori bound instead of the tracked |max|), i.e. last-bit
    differences.  Strips of 60 columns: widths around the strip boundary (60, 61), one strip narrower than a block, a
    one-pixel image, the real stage-1 geometry (80 x 798).  Borders of the output stay exactly zero."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(H * 1000 + W + 7)
    x = torch.randn(B, 32, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, 1, 1, 1, generator=g))
    w1 = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    w2 = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    b1, b2 = torch.randn(32, generator=g) * 0.3, torch.randn(32, generator=g) * 0.3

    def padded(t):
        o = torch.zeros(B, H + 2, W + 2, 32)
        o[:, 1:-1, 1:-1] = t.permute(0, 2, 3, 1)
        return o.contiguous()
    wp1 = w1.permute(0, 2, 3, 1).reshape(32, 288).contiguous()          # k = (dh*3 + dw)*32 + ci
    wp2 = w2.permute(0, 2, 3, 1).reshape(32, 288).contiguous()
    h1, h2 = ops.split_weights_h2(wp1.to(gpu)), ops.split_weights_h2(wp2.to(gpu))
    W31, W32 = ops.split_weights(wp1.to(gpu)), ops.split_weights(wp2.to(gpu))
    xin = padded(x).to(gpu)
    fused = ops.resblock32_fused(xin, h1, b1.to(gpu), h2, b2.to(gpu), float(wp1.abs().sum(1).max()) * (1 + 1e-6),
                                 float(b1.abs().max())).cpu()
    mid = ops.conv3x3_c32(xin, W31, b1.to(gpu), relu=True, h2_weights=h1)
    two = ops.conv3x3_c32(mid, W32, b2.to(gpu), R=xin, post_relu=True, h2_weights=h2).cpu()
    F = torch.nn.functional
    ref = torch.relu(F.conv2d(torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(),
                              padding=1) + x.double())
    got = fused[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double()
    assert _rel_err(got, ref) < 1e-5
    scale = two.abs().max().item()
    assert (fused - two).abs().max().item() <= 2e-6 * scale, ((fused - two).abs().max().item(), scale)
    assert fused[:, 0].abs().max() == 0 and fused[:, -1].abs().max() == 0
    assert fused[:, :, 0].abs().max() == 0 and fused[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("C,H,W,B", [(32, 5, 37, 2), (32, 80, 798, 1), (32, 7, 61, 2), (32, 1, 1, 1), (64, 5, 37, 2), (64, 40, 399, 2),
                                     (64, 3, 60, 1), (64, 7, 121, 3), (64, 2, 1, 1)])
def test_resblock_ws_matches_float64_and_the_fused_form(built_lib, gpu, C, H, W, B):
    """(r4) csrc/resblock_ws.hip — the BasicBlock with producer / consumer wavefronts (conv1 and conv2 run concurrently, one
    row apart, through LDS line buffers), C = 32 and C = 64 planes (ResNet stages 1 and 2) — against the block evaluated by
    torch in float64; for C = 32 also against resblock_fused.hip, which performs the same arithmetic in alternating phases
    (identical bits expected: same planes, same scales, same product order)."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(C * 100000 + H * 1000 + W)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, 1, 1, 1, generator=g))
    w1 = torch.randn(C, C, 3, 3, generator=g) * (0.08 * (32 / C) ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) * (0.08 * (32 / C) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3

    def padded(t):
        o = torch.zeros(B, H + 2, W + 2, C)
        o[:, 1:-1, 1:-1] = t.permute(0, 2, 3, 1)
        return o.contiguous()
    wp1 = w1.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()          # k = (dh*3 + dw)*C + ci
    wp2 = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    h1, h2 = ops.split_weights_h2(wp1.to(gpu)), ops.split_weights_h2(wp2.to(gpu))
    xin = padded(x).to(gpu)
    l1, bm = float(wp1.abs().sum(1).max()) * (1 + 1e-6), float(b1.abs().max())
    out = ops.resblock_ws(xin, h1, b1.to(gpu), h2, b2.to(gpu), l1, bm).cpu()
    F = torch.nn.functional
    ref = torch.relu(F.conv2d(torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(),
                              padding=1) + x.double())
    assert _rel_err(out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), ref) < 1e-5
    assert out[:, 0].abs().max() == 0 and out[:, -1].abs().max() == 0
    assert out[:, :, 0].abs().max() == 0 and out[:, :, -1].abs().max() == 0
    if C == 32:
        fused = ops.resblock32_fused(xin, h1, b1.to(gpu), h2, b2.to(gpu), l1, bm).cpu()
        assert torch.equal(out, fused)


@pytest.mark.parametrize("C,H,W,B", [(32, 7, 61, 2), (32, 80, 798, 1), (64, 5, 37, 2), (64, 40, 399, 2), (64, 2, 1, 1)])
def test_resblock_single_term_forms(built_lib, gpu, C, H, W, B):
    """(r5) the fused BasicBlock kernels with ONE fp16 term per operand (DZN_PREC_F16: resblock32_fused_kernel<1>,
    resblock_ws_kernel<C, 1>, via the dzn_op_set_resblock_np test knob): against the block in float64 the error must sit at the
    fp16 rounding level of two chained convolutions (a few 2^-11 of the block's |max| — the embedding trunk is three orders of
    magnitude inside its cosine bar at that level), far above the two-term form's 1e-5 and far below a broken plane stride;
    for C = 32 the two kernels give identical bits; borders stay zero."""
    from diarizen_amd import _lib, ops
    g = torch.Generator().manual_seed(C * 100000 + H * 1000 + W + 7)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, 1, 1, 1, generator=g))
    w1 = torch.randn(C, C, 3, 3, generator=g) * (0.08 * (32 / C) ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) * (0.08 * (32 / C) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3

    def padded(t):
        o = torch.zeros(B, H + 2, W + 2, C)
        o[:, 1:-1, 1:-1] = t.permute(0, 2, 3, 1)
        return o.contiguous()
    wp1 = w1.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    wp2 = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    h1, h2 = ops.split_weights_h2(wp1.to(gpu)), ops.split_weights_h2(wp2.to(gpu))
    xin = padded(x).to(gpu)
    l1, bm = float(wp1.abs().sum(1).max()) * (1 + 1e-6), float(b1.abs().max())
    lib = _lib.load()
    two = ops.resblock_ws(xin, h1, b1.to(gpu), h2, b2.to(gpu), l1, bm).cpu()
    assert lib.dzn_op_set_resblock_np(1) == 0
    try:
        one = ops.resblock_ws(xin, h1, b1.to(gpu), h2, b2.to(gpu), l1, bm).cpu()
        fused = ops.resblock32_fused(xin, h1, b1.to(gpu), h2, b2.to(gpu), l1, bm).cpu() if C == 32 else None
    finally:
        lib.dzn_op_set_resblock_np(2)
    F = torch.nn.functional
    ref = torch.relu(F.conv2d(torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(),
                              padding=1) + x.double())
    e1 = _rel_err(one[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), ref)
    e2 = _rel_err(two[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), ref)
    print(f"[resblock C={C} {H}x{W}] single term {e1:.2e}, two terms {e2:.2e}")
    assert e2 < 1e-5 and 1e-5 < e1 < 4e-3
    assert one[:, 0].abs().max() == 0 and one[:, -1].abs().max() == 0 and one[:, :, 0].abs().max() == 0 and one[:, :, -1].abs().max() == 0
    if fused is not None:
        assert torch.equal(one, fused)


@pytest.mark.parametrize("M,N,K", [(300, 64, 256), (1000, 200, 1024), (257, 1024, 96)])
def test_gemm_presplit_operand(built_lib, gpu, M, N, K):
    """gemm_split_pre.hip: A handed over as three bf16 planes (dzn_op_split_rows) gives the same result as the
    in-kernel split, to fp32 round-off, for narrow (128x64 tiles) and wide (256x128 tiles) outputs."""
    from diarizen_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g))
    W = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    planes = ops.split_rows(A.to(gpu))
    rec = planes.cpu().view(torch.bfloat16).float().sum(0)                 # hi + mid + lo, fragment order
    pos = torch.tensor([8 * ((k & 15) >> 2) + (k & 3) + 4 * (k >> 4) for k in range(32)])
    assert torch.equal(rec.view(M, K // 32, 32)[:, :, pos].reshape(M, K), A)
    out = ops.gemm(A.to(gpu), W.to(gpu), bias=bias.to(gpu), precision=2, a_planes=planes)
    assert _rel_err(out.cpu(), ref) < 1e-5
