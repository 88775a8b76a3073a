"""Thin torch-tensor wrappers over the kernel-level C ABI (include/dzn_ops.h).

Only used by tests/ and bench.py to exercise single kernels; the engine itself
orchestrates the same launchers in C++.  Tensors must live on the HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import DznGemmDesc, check


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def gemm(A, W, *, M=None, N=None, K=None, bias=None, R=None, C_out=None, WS=None, ws_w=0.0,
         ws_init=False, a_rowoff=None, c_rowoff=None, lda=None, kc=0, ldk=0, ldw=None, ldc=None,
         ldws=0, act=0, alpha=1.0, post_relu=False, nz=1, zdiv=1, zs=None, precision=0,
         W16=None, W3=None, a_planes=None, ln_stats=None, ln_colsum=None, W2h=None, col_scale=None,
         a_amax=None, c_amax=None, amax_unit=None, want_row_stats=False, stat_eps=1e-5, mx=False, Wmx=None,
         col_scale_mx=None, kv_col0=None):
    """C = epilogue(A @ W^T); see dzn_gemm_desc.  kv_col0 (r6): columns >= kv_col0 leave as fp16 two-term planes with per-(row,
    64-column slot) scales instead of fp32 (dzn_gemm_desc.kv_planes) -> returns (C, planes int16 [2, M, N - kv_col0], inv f32
    [M, (N - kv_col0) / 64]).  A: [M, K] (or raw buffer with lda / rowoff),
    W: [N, K] fp32 (and optionally W16 bf16)."""
    lib = _lib.load()
    assert A.is_cuda and A.dtype in (torch.float32, torch.bfloat16)
    a_bf16 = A.dtype == torch.bfloat16
    if a_bf16 and C_out is None:
        raise ValueError("bf16 A: pass C_out (fp32 or bf16) explicitly")
    if N is None:
        N = W.shape[0]
    if K is None:
        K = W.shape[1]
    if M is None:
        M = A.shape[0]
    if lda is None:
        lda = A.stride(0) if A.dim() == 2 else K
    if ldw is None:
        ldw = W.stride(0) if W is not None else W16.stride(0)
    if C_out is None:
        C_out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    if ldc is None:
        ldc = C_out.stride(0) if C_out.dim() == 2 else N
    if precision in (_lib.DZN_PREC_F32_SPLIT, _lib.DZN_PREC_F32_H2, _lib.DZN_PREC_F16) and W3 is None and K % 32 == 0 and ldw == K and W.is_contiguous():
        W3 = split_weights(W.reshape(-1, K))   # convenience for tests: engines split once at load
    d = DznGemmDesc()
    d.A, d.W, d.W16, d.C = _p(A), _p(W), _p(W16), _p(C_out)
    d.bias, d.R, d.WS = _p(bias), _p(R), _p(WS)
    d.a_rowoff, d.c_rowoff = _p(a_rowoff), _p(c_rowoff)
    d.M, d.N, d.K = M, N, K
    d.lda, d.kc, d.ldk, d.ldw, d.ldc, d.ldws = lda, kc, ldk, ldw, ldc, ldws
    d.act, d.alpha, d.post_relu = act, alpha, int(post_relu)
    d.ws_w, d.ws_init = ws_w, int(ws_init)
    d.nz, d.zdiv = nz, zdiv
    if zs:
        for k, v in zs.items():
            setattr(d, k, v)
    d.precision = precision
    d.a_bf16 = int(a_bf16)
    d.c_bf16 = int(C_out.dtype == torch.bfloat16)
    d.r_bf16 = int(R is not None and R.dtype == torch.bfloat16)
    d.W3 = _p(W3)
    if a_planes is not None:      # A pre-split by split_rows(): [3, M, K] int16 planes
        d.A = _p(a_planes)
        d.a_split3, d.a_plane = 1, a_planes.stride(0)
    d.ln_stats, d.ln_colsum = _p(ln_stats), _p(ln_colsum)
    if precision in (_lib.DZN_PREC_F32_H2, _lib.DZN_PREC_F16) and W2h is None and K % 32 == 0 and ldw == K and W.is_contiguous():
        W2h, col_scale = split_weights_h2(W.reshape(-1, K))
        if a_amax is None:      # one |max| for the whole tensor, replicated for every z scale unit
            a_amax = amax(A).repeat(max(1, nz // max(zdiv, 1)))
    if mx and Wmx is None:          # DZN_PREC_F16 with fp8 cross terms (csrc/gemm_mx.hip)
        assert precision == _lib.DZN_PREC_F16 and K % 32 == 0 and ldw == K and W.is_contiguous()
        Wmx, col_scale_mx = split_weights_mx(W.reshape(-1, K))
    d.W2h, d.col_scale, d.a_amax, d.c_amax = _p(W2h), _p(col_scale), _p(a_amax), _p(c_amax)
    d.Wmx, d.col_scale_mx = _p(Wmx), _p(col_scale_mx)
    # one scale unit for the whole tensor unless told otherwise (engines use one unit per window)
    d.amax_unit = int(amax_unit) if amax_unit is not None else (max(M, 1) if nz == 1 else 0)
    stats = None
    if want_row_stats:      # LayerNorm statistics of the output rows, left by the epilogue + finalize kernel
        part = torch.empty((M, 32, 2), device=A.device, dtype=torch.float32)
        stats = torch.empty((M, 2), device=A.device, dtype=torch.float32)
        d.stat_partial, d.stat_final, d.stat_C, d.stat_eps = _p(part), _p(stats), N, stat_eps
    kvp = kvs = None
    if kv_col0 is not None:
        kvp = torch.zeros((2, M, N - kv_col0), device=A.device, dtype=torch.int16)
        kvs = torch.zeros((M, (N - kv_col0) // 64), device=A.device, dtype=torch.float32)
        d.kv_planes, d.kv_plane_stride, d.kv_scale, d.kv_ld, d.kv_col0 = _p(kvp), kvp.stride(0), _p(kvs), N - kv_col0, kv_col0
    check(lib.dzn_op_gemm(C.byref(d), _stream()), what="dzn_op_gemm")
    if kv_col0 is not None:
        return C_out, kvp, kvs
    return (C_out, stats) if want_row_stats else C_out


def attention_planes(qkv, B, L, h, gate=None, table=None, head_idx=None, Htot=0, scale=0.125):
    """(r6) csrc/attention_planes.hip through its test entry point: the K / V slots of `qkv` are packed into fp16 two-term planes
    with per-(row, head) power-of-two scales (as the q/k/v contraction's epilogue does) and the planes kernel runs on them."""
    lib = _lib.load()
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.shape == (B * L, 3 * h * 64)
    out = torch.empty((B * L, h * 64), device=qkv.device, dtype=torch.float32)
    am = qkv.reshape(B, -1).abs().amax(dim=1).float().contiguous()
    planes = torch.zeros((2, B * L + 64, 2 * h * 64), device=qkv.device, dtype=torch.int16)
    kvs = torch.zeros((B * L + 64, 2 * h), device=qkv.device, dtype=torch.float32)
    check(lib.dzn_op_attention_planes(_p(qkv), _p(out), _p(gate), _p(table), _p(head_idx), B, L, h, Htot, qkv.stride(0),
                                      out.stride(0), scale, _p(am), _p(planes), _p(kvs), _stream()), what="dzn_op_attention_planes")
    return out


def split_weights(W):
    """Exact 3-way bf16 split of fp32 weights [rows, K] (K % 32 == 0) for precision=DZN_PREC_F32_SPLIT:
    returns the packed planes as an int16 tensor [rows, K // 32, 3, 32] (csrc/gemm_split.hip)."""
    lib = _lib.load()
    assert W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.shape[1] % 32 == 0
    rows, K = W.shape
    out = torch.empty((rows, K // 32, 3, 32), device=W.device, dtype=torch.int16)
    check(lib.dzn_op_split_weights(_p(W), rows, K, W.stride(0), _p(out), _stream()),
          what="dzn_op_split_weights")
    return out


def split_weights_h2(W):
    """two-term fp16 split for precision=DZN_PREC_F32_H2: (planes int16 [rows, K // 32, 2, 32], col_scale f32 [rows])"""
    lib = _lib.load()
    assert W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.shape[1] % 32 == 0
    rows, K = W.shape
    out = torch.empty((rows, K // 32, 2, 32), device=W.device, dtype=torch.int16)
    sc = torch.empty((rows,), device=W.device, dtype=torch.float32)
    check(lib.dzn_op_split_weights_h2(_p(W), rows, K, W.stride(0), _p(out), _p(sc), _stream()),
          what="dzn_op_split_weights_h2")
    return out, sc


def split_weights_mx(W):
    """planes of the reduced-precision contraction (DZN_PREC_F16, csrc/gemm_mx.hip): (uint8 [rows, K // 32, 128], col_scale f32 [rows])"""
    lib = _lib.load()
    assert W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.shape[1] % 32 == 0
    rows, K = W.shape
    out = torch.empty((rows, K // 32, 128), device=W.device, dtype=torch.uint8)
    sc = torch.empty((rows,), device=W.device, dtype=torch.float32)
    check(lib.dzn_op_split_weights_mx(_p(W), rows, K, W.stride(0), _p(out), _p(sc), _stream()),
          what="dzn_op_split_weights_mx")
    return out, sc


def amax(x, out=None):
    """device scalar max(out, max |x|) — the tracker a producer without a fused one would keep"""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.zeros((1,), device=x.device, dtype=torch.float32)
    check(lib.dzn_op_amax(_p(x), x.numel(), _p(out), _stream()), what="dzn_op_amax")
    return out


def split_rows(x):
    """fp32 [rows, D] -> int16 [3, rows, D]: the three bf16 planes of the exact split, fragment order."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] % 32 == 0
    rows, D = x.shape
    out = torch.empty((3, rows, D), device=x.device, dtype=torch.int16)
    check(lib.dzn_op_split_rows(_p(x), _p(out), out.stride(0), rows, D, _stream()), what="dzn_op_split_rows")
    return out


def conv3x3_c32(img, W3, bias, R=None, relu=False, post_relu=False, out=None, h2_weights=None):
    """3x3 stride-1 conv 32 -> 32 on zero-bordered NHWC fp32 images [B, H+2, W+2, 32] (csrc/conv_split.hip);
    W3 = split_weights(W[32, 288]) with k = (dh*3 + dw)*32 + ci.  Returns a new zero-bordered image."""
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and img.shape[-1] == 32
    B, Hp, Wp, _ = img.shape
    if out is None:
        out = torch.zeros_like(img)
    if h2_weights is not None:      # fp16 two-term variant: (W2h, col_scale) + per-image |max| of the input
        W2h, cs = h2_weights
        am = img.reshape(B, -1).abs().amax(dim=1).float().contiguous()
        check(lib.dzn_op_conv3x3_c32_h2(_p(img), _p(W3), _p(W2h), _p(cs), _p(am), _p(bias), _p(R), _p(out), B, Hp - 2,
                                        Wp - 2, int(relu), int(post_relu), _stream()), what="dzn_op_conv3x3_c32_h2")
        return out
    check(lib.dzn_op_conv3x3_c32(_p(img), _p(W3), _p(bias), _p(R), _p(out), B, Hp - 2, Wp - 2, int(relu),
                                 int(post_relu), _stream()), what="dzn_op_conv3x3_c32")
    return out


def resblock32_fused(img, h2_w1, b1, h2_w2, b2, l1max1: float, bmax1: float, out=None):
    """one 32-channel BasicBlock, out = relu(conv2(relu(conv1(img) + b1)) + b2 + img), in ONE kernel
    (csrc/resblock_fused.hip): zero-bordered NHWC fp32 images [B, H+2, W+2, 32]; h2_w = split_weights_h2(W[32, 288])
    = (planes, inverse row scales); l1max1 / bmax1 bound the intermediate: max_oc sum_k |W1[oc][k]|, max_oc |b1[oc]|."""
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and img.shape[-1] == 32
    B, Hp, Wp, _ = img.shape
    if out is None:
        out = torch.zeros_like(img)
    am = img.reshape(B, -1).abs().amax(dim=1).float().contiguous()
    (W1, cs1), (W2, cs2) = h2_w1, h2_w2
    check(lib.dzn_op_resblock32_fused(_p(img), _p(out), _p(W1), _p(cs1), _p(b1), _p(W2), _p(cs2), _p(b2), _p(am),
                                      float(l1max1), float(bmax1), B, Hp - 2, Wp - 2, _stream()),
          what="dzn_op_resblock32_fused")
    return out


def resblock_ws(img, h2_w1, b1, h2_w2, b2, l1max1: float, bmax1: float, out=None):
    """the same block with producer / consumer wavefronts (csrc/resblock_ws.hip), C = img.shape[-1] in (32, 64);
    h2_w = split_weights_h2(W[C, 9 C]) with k = (dh*3 + dw) * C + ci"""
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and img.shape[-1] in (32, 64)
    B, Hp, Wp, Cn = img.shape
    if out is None:
        out = torch.zeros_like(img)
    am = img.reshape(B, -1).abs().amax(dim=1).float().contiguous()
    (W1, cs1), (W2, cs2) = h2_w1, h2_w2
    check(lib.dzn_op_resblock_ws(_p(img), _p(out), _p(W1), _p(cs1), _p(b1), _p(W2), _p(cs2), _p(b2), _p(am), float(l1max1),
                                 float(bmax1), B, Hp - 2, Wp - 2, Cn, _stream()), what="dzn_op_resblock_ws")
    return out


def linkage_centroid(emb, device: int = -1):
    """scipy.cluster.hierarchy.linkage(emb, "centroid", "euclidean") on the device (csrc/linkage.hip):
    emb = host float32 [n, dim] (numpy), returns the dendrogram float64 [n - 1, 4]."""
    import numpy as np
    lib = _lib.load()
    e = np.ascontiguousarray(emb, dtype=np.float32)
    n, dim = e.shape
    Z = np.empty((n - 1, 4), dtype=np.float64)
    check(lib.dzn_linkage_centroid(e.ctypes.data_as(C.c_void_p), n, dim, Z.ctypes.data_as(C.c_void_p), device),
          what="dzn_linkage_centroid")
    return Z


def cdist_cosine(emb, centroids, device: int = -1):
    """scipy.spatial.distance.cdist(emb, centroids, "cosine") on the device in scipy's float64 operation order
    (csrc/linkage.hip): emb = host float32 [n, dim], centroids [k, dim] -> float64 [n, k]."""
    import numpy as np
    lib = _lib.load()
    e = np.ascontiguousarray(emb, dtype=np.float32)
    c = np.ascontiguousarray(centroids, dtype=np.float64)
    n, dim = e.shape
    k = c.shape[0]
    assert c.shape[1] == dim
    out = np.empty((n, k), dtype=np.float64)
    check(lib.dzn_cdist_cosine(e.ctypes.data_as(C.c_void_p), n, dim, c.ctypes.data_as(C.c_void_p), k,
                               out.ctypes.data_as(C.c_void_p), device), what="dzn_cdist_cosine")
    return out


class VbxState:
    """the E-sized arrays of the VBx mixture resident on the device (csrc/vbx.hip): X f64 [E, D], Phi f64 [D],
    gamma0 f64 [E, K]; see diarizen_amd/clustering.py:vb_gmm for the loop that drives it."""

    def __init__(self, X, Phi, gamma0, device: int = -1):
        import numpy as np
        self.np, self.lib = np, _lib.load()
        X = np.ascontiguousarray(X, dtype=np.float64)
        Phi = np.ascontiguousarray(Phi, dtype=np.float64)
        g0 = np.ascontiguousarray(gamma0, dtype=np.float64)
        self.E, self.D = X.shape
        self.K = g0.shape[1]
        assert Phi.shape == (self.D,) and g0.shape[0] == self.E
        self.state = C.c_void_p()
        check(self.lib.dzn_vbx_create(X.ctypes.data_as(C.c_void_p), Phi.ctypes.data_as(C.c_void_p),
                                      g0.ctypes.data_as(C.c_void_p), self.E, self.D, self.K, device,
                                      C.byref(self.state)), what="dzn_vbx_create")

    def stats(self):
        out = self.np.empty((self.K, self.D + 1), dtype=self.np.float64)
        check(self.lib.dzn_vbx_stats(self.state, out.ctypes.data_as(C.c_void_p)), what="dzn_vbx_stats")
        return out

    def estep(self, alpha, ck, lpi, Fa: float) -> float:
        np = self.np
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        ck = np.ascontiguousarray(ck, dtype=np.float64)
        lpi = np.ascontiguousarray(lpi, dtype=np.float64)
        assert alpha.shape == (self.K, self.D) and ck.shape == (self.K,) and lpi.shape == (self.K,)
        total = C.c_double(0.0)
        check(self.lib.dzn_vbx_estep(self.state, alpha.ctypes.data_as(C.c_void_p), ck.ctypes.data_as(C.c_void_p),
                                     lpi.ctypes.data_as(C.c_void_p), float(Fa), C.byref(total)), what="dzn_vbx_estep")
        return total.value

    def gamma(self):
        out = self.np.empty((self.E, self.K), dtype=self.np.float64)
        check(self.lib.dzn_vbx_gamma(self.state, out.ctypes.data_as(C.c_void_p)), what="dzn_vbx_gamma")
        return out

    def close(self):
        if self.state:
            self.lib.dzn_vbx_destroy(self.state)
            self.state = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # pragma: no cover
            pass


def layernorm(x, gamma, beta, C_true=None, eps=1e-5, gelu=False, out=None):
    lib = _lib.load()
    rows, ld = x.shape[0], x.stride(0)
    Cpad = x.shape[1]
    if C_true is None:
        C_true = Cpad
    if out is None:
        out = torch.empty_like(x)
    check(lib.dzn_op_layernorm(_p(x), ld, _p(out), out.stride(0), _p(gamma), _p(beta), rows,
                               C_true, Cpad, eps, int(gelu), _stream()), what="dzn_op_layernorm")
    return out


def row_stats(x, C_true=None, eps=1e-5):
    """(mean, rstd) per row of x[:, :C_true] -> f32 [rows, 2] (the LayerNorm folded into a contraction)"""
    lib = _lib.load()
    rows = x.shape[0]
    out = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    check(lib.dzn_op_row_stats(_p(x), x.stride(0), rows, C_true or x.shape[1], eps, _p(out), _stream()),
          what="dzn_op_row_stats")
    return out


def gate_stats(x, gamma, beta, Wg, bg, cst, eps=1e-5):
    """one pass over raw rows x [rows, Htot*64]: (gate [rows, Htot] on LayerNorm(x), stats [rows, 2])"""
    lib = _lib.load()
    rows, Htot = x.shape[0], cst.numel()
    gate_ = torch.empty((rows, Htot), device=x.device, dtype=torch.float32)
    stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    check(lib.dzn_op_gate_stats(_p(x), x.stride(0), _p(gamma), _p(beta), _p(Wg), _p(bg), _p(cst), _p(gate_), _p(stats),
                                rows, Htot, eps, _stream()), what="dzn_op_gate_stats")
    return gate_, stats


def gate(y, Wg, bg, cst):
    lib = _lib.load()
    rows = y.shape[0]
    Htot = cst.numel()
    out = torch.empty((rows, Htot), device=y.device, dtype=torch.float32)
    check(lib.dzn_op_gate(_p(y), y.stride(0), _p(Wg), _p(bg), _p(cst), _p(out), rows, Htot,
                          _stream()), what="dzn_op_gate")
    return out


def attention(qkv, B, L, h, *, gate=None, table=None, head_idx=None, Htot=0, scale=0.125,
              precision=0):
    lib = _lib.load()
    out = torch.empty((B * L, h * 64), device=qkv.device, dtype=torch.float32)
    if precision == _lib.DZN_PREC_F32_H2:      # fp16 two-term variant: per-window |max| of qkv
        am = qkv.reshape(B, -1).abs().amax(dim=1).float().contiguous()
        check(lib.dzn_op_attention_h2(_p(qkv), _p(out), _p(gate), _p(table), _p(head_idx), B, L, h, Htot,
                                      qkv.stride(0), out.stride(0), scale, _p(am), _stream()), what="dzn_op_attention_h2")
        return out
    check(lib.dzn_op_attention(_p(qkv), _p(out), _p(gate), _p(table), _p(head_idx), B, L, h,
                               Htot, qkv.stride(0), out.stride(0), scale, precision, _stream()),
          what="dzn_op_attention")
    return out
