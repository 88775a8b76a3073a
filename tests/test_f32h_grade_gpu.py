"""Is the two-term fp16 split ("f32h": 22 significant bits per operand, 3 MFMA products) fp32-grade on EVERY contraction
shape of the headline step — not only on the one long-K shape of tests/test_ops_gpu.py::test_gemm_split_is_fp32_grade?

For each distinct (N, K) of the 30-min wavlm-large-s80 step (profiles/r2_kernel_shapes_f32h_30min_b384.txt; K = 32 ... 8192,
where for short K the 2^-22 operand truncation, not accumulation rounding, is the larger term) the three fp32 modes run on
the same operands against a float64 product; the error is measured relative to sum_k |a||w| (the scale fp32 rounding lives on).

  * wide-range data (activations spanning ~5 decades inside a scale unit): f32h <= 1.5 x the hardware fp32 MFMA's error,
    max and rms, per shape;
  * the adversarial tensor of split.h's header — one element per scale unit at the unit's |max|, the rest 2^-17 ... 2^-20
    below it, so that every lo term is an fp16 SUBNORMAL: the documented floor applies — absolute error per A element
    <= 2^-39 |max| (2^-25 in scaled units), i.e. |err| <= 1.5 x fp32-MFMA + 2^-38 |max| sum_k |w|.  The measured ratios
    go to gpurun_out/f32h_grade.json (copied to profiles/ by scripts/final_measure.sh).
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 32), (64, 288), (64, 576), (64, 8192), (96, 1024), (128, 64), (128, 576), (128, 1024), (128, 1152), (160, 1024),
          (192, 1024), (224, 480), (224, 768), (224, 1024), (256, 128), (256, 256), (256, 672), (256, 1024), (256, 1152),
          (256, 2304), (288, 1024), (320, 768), (320, 1024), (384, 640), (384, 1024), (416, 1024), (480, 1024), (512, 256),
          (544, 1024), (576, 1024), (640, 1024), (672, 1024), (704, 1024), (768, 256), (768, 1024), (928, 1024), (960, 1024),
          (1024, 64), (1024, 96), (1024, 128), (1024, 160), (1024, 192), (1024, 224), (1024, 256), (1024, 288), (1024, 320),
          (1024, 384), (1024, 416), (1024, 480), (1024, 512), (1024, 544), (1024, 576), (1024, 640), (1024, 672), (1024, 704),
          (1024, 768), (1024, 928), (1024, 960), (1024, 1024), (1024, 1120), (1024, 1344), (1024, 1792), (1120, 1024),
          (1152, 1024), (1344, 1024), (1536, 1024), (1728, 1024), (1792, 1024), (1920, 1024)]
UNIT, UNITS = 399, 2            # rows per scale unit (one 8 s window), units per test tensor
_REPORT = {}


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    if _REPORT:
        os.makedirs("gpurun_out", exist_ok=True)
        worst = {k: max(v[k] for v in _REPORT.values()) for k in next(iter(_REPORT.values()))}
        json.dump({"per_shape": _REPORT, "worst": worst, "unit_rows": UNIT,
                   "note": "error vs float64 relative to sum|a||w|; ratios are f32h / fp32-MFMA"},
                  open("gpurun_out/f32h_grade.json", "w"), indent=1)


def _errors(A, W, gpu):
    from diarizen_amd import ops
    M = A.shape[0]
    ref = A.double() @ W.double().T
    scale = A.double().abs() @ W.double().abs().T
    Ag, Wg = A.to(gpu), W.to(gpu)
    am = torch.stack([Ag[u * UNIT:(u + 1) * UNIT].abs().max() for u in range(UNITS)]).contiguous()
    out = {}
    for prec in (0, 2, 3):
        kw = dict(a_amax=am, amax_unit=UNIT) if prec == 3 else {}
        if prec == 3:
            kw["W2h"], kw["col_scale"] = ops.split_weights_h2(Wg)
        c = ops.gemm(Ag, Wg, precision=prec, **kw).cpu().double()
        e = (c - ref).abs()
        out[prec] = (e, e / scale)
    return out, ref, scale


@pytest.mark.parametrize("N,K", SHAPES)
def test_f32h_is_fp32_grade_per_shape(built_lib, gpu, N, K):
    g = torch.Generator().manual_seed(1000 * N + K)
    M = UNIT * UNITS
    A = torch.randn(M, K, generator=g) * torch.exp(2.0 * torch.randn(M, K, generator=g))
    A[UNIT:] *= 37.0                                   # the two units get different power-of-two scales
    W = torch.randn(N, K, generator=g) * 0.05 * torch.exp(0.5 * torch.randn(N, 1, generator=g))
    errs, _, _ = _errors(A, W, gpu)
    mx = {p: errs[p][1].max().item() for p in errs}
    rms = {p: errs[p][1].pow(2).mean().sqrt().item() for p in errs}
    _REPORT.setdefault(f"N{N}_K{K}", {}).update(wide_max_ratio=mx[3] / mx[0], wide_rms_ratio=rms[3] / rms[0],
                                                wide_f32s_max_ratio=mx[2] / mx[0], wide_f32_max=mx[0])
    assert mx[3] <= 1.5 * mx[0] and rms[3] <= 1.5 * rms[0], (N, K, mx, rms)
    assert mx[2] <= 1.5 * mx[0] and rms[2] <= 1.5 * rms[0], (N, K, mx, rms)


@pytest.mark.parametrize("N,K", SHAPES)
def test_f32h_subnormal_lo_regime_meets_documented_floor(built_lib, gpu, N, K):
    g = torch.Generator().manual_seed(7000 * N + K)
    M = UNIT * UNITS
    # everything 2^-17 ... 2^-20 below the unit's maximum (random mantissas, random signs) ...
    expo = torch.randint(17, 21, (M, K), generator=g).float()
    A = (1.0 + torch.rand(M, K, generator=g)) * torch.exp2(-expo) * torch.where(torch.rand(M, K, generator=g) < 0.5, -1.0, 1.0)
    amax = [3.0, 700.0]
    for u in range(UNITS):
        A[u * UNIT:(u + 1) * UNIT] *= amax[u]
        A[u * UNIT + 5, K // 3] = amax[u]             # ... and ONE element at the maximum
    W = torch.randn(N, K, generator=g) * 0.05
    errs, _, scale = _errors(A, W, gpu)
    unit_max = torch.tensor(amax).repeat_interleave(UNIT).double()[:, None]
    floor = 2.0 ** -38 * unit_max * W.double().abs().sum(1)[None, :]
    e3, e0 = errs[3][0], errs[0][0]
    bound = 1.5 * e0.max() + floor                      # per output: fp32-MFMA level + the documented subnormal-lo floor
    ratio = (errs[3][1].max() / errs[0][1].max()).item()
    _REPORT.setdefault(f"N{N}_K{K}", {}).update(adversarial_max_ratio=ratio,
                                                adversarial_err_over_floor=((e3 - 1.5 * e0.max()).clamp(min=0) / floor).max().item())
    assert (e3 <= bound).all(), (N, K, ratio)
    # f32s is exact in its operands: it must stay at the fp32 MFMA's level here too
    assert errs[2][1].max() <= 1.5 * errs[0][1].max(), (N, K)
