"""CPU model of the arithmetic of diarizen_amd/csrc/gemm_mx.hip (test infrastructure; torch only, no GPU).

    C = hi16(A s_a) hi16(W s_w)^T + [ fp8(A s_a) fp8(lo_w 2^11)^T + fp8(lo_a 2^11) fp8(W s_w)^T ] 2^-11,   then / (s_a s_w)

with s_a = 2^(7 - floor(log2 amax_a)) per scale unit (window), s_w the same per weight row, lo = x s - hi16(x s) (exact),
fp8 = OCP e4m3fn, round to nearest even (torch.float8_e4m3fn).  Products are exact in float64 here; the device accumulates in
fp32, so the two agree to fp32-accumulation level when the device's operand layout, block scales and conversions are right.
"""
from __future__ import annotations

import torch


def _pow2_scale(amax: torch.Tensor) -> torch.Tensor:
    e = torch.floor(torch.log2(amax.double().clamp_min(1e-300)))
    return torch.exp2(7.0 - e)


def mx_terms(x: torch.Tensor, scale: torch.Tensor):
    """x float32 [rows, K], scale float64 broadcastable: (hi16, hi8, lo8 * 2^-11) as float64, all in SCALED units"""
    xs = (x.double() * scale).float()                      # exact: power-of-two scaling of fp32 values
    hi = xs.to(torch.float16).float()
    lo = (xs - hi) * 2048.0                                  # exact in fp32
    hi8 = xs.to(torch.float8_e4m3fn).float()
    lo8 = lo.to(torch.float8_e4m3fn).float()
    return hi.double(), hi8.double(), lo8.double() / 2048.0


def mx_gemm(A: torch.Tensor, W: torch.Tensor, a_amax: torch.Tensor | float | None = None) -> torch.Tensor:
    """float64 [M, N] = the value gemm_mx.hip computes for A [M, K], W [N, K] (before bias / epilogue) with exact accumulation.
    a_amax: the |max| bound the kernel is given for A (one unit) — default max |A|."""
    amax = A.abs().max() if a_amax is None else torch.as_tensor(a_amax)
    sa = _pow2_scale(amax)
    sw = _pow2_scale(W.abs().amax(dim=1, keepdim=True))
    ah, ah8, al8 = mx_terms(A, sa)
    wh, wh8, wl8 = mx_terms(W, sw)
    acc = ah @ wh.T + ah8 @ wl8.T + al8 @ wh8.T
    return acc / sa / sw.T


def single_term_gemm(A: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """hi16 hi16 only (r2-r4's DZN_PREC_F16 arithmetic), for scale"""
    sa = _pow2_scale(A.abs().max())
    sw = _pow2_scale(W.abs().amax(dim=1, keepdim=True))
    ah = (A.double() * sa).float().to(torch.float16).double()
    wh = (W.double() * sw).float().to(torch.float16).double()
    return (ah @ wh.T) / sa / sw.T
