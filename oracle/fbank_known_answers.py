"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

Hand-derived known answers for the kaldi filter bank (PA/models/embedding/wespeaker/__init__.py:80-103 ->
torchaudio.compliance.kaldi.fbank, absent offline): three 400-sample frames whose log-mel energies have CLOSED FORMS, so
that the two restatements (oracle/emb_model.py:kaldi_fbank after torchaudio's call graph, oracle/kaldi_fbank_ref.py after
Kaldi's C++) and the device kernels are held to something that is neither of them and contains no FFT, no matrix product
and no loop over samples.  Pure Python (math / cmath, float64).

    python oracle/fbank_known_answers.py        # rewrites tests/golden/fbank_known_answers.json

Notation: N = 400 samples per frame, P = 512 FFT points, theta_k = 2 pi k / P, Hamming h[n] = 0.54 - 0.46 cos(a n) with
a = 2 pi / (N - 1), pre-emphasis c = 0.97 applied as y[n] = u[n] - c u[n-1] (n >= 1), y[0] = (1 - c) u[0] after the DC
removal u = x - mean(x).

  D(phi)  = sum_{n=0}^{N-1} e^{-i phi n} = (1 - e^{-i N phi}) / (1 - e^{-i phi})            (N when phi = 0 mod 2 pi)
  Hh(phi) = sum_n h[n] e^{-i phi n} = 0.54 D(phi) - 0.23 (D(phi - a) + D(phi + a))

  frame "dc"       x[n] = C.  u = 0, so every mel energy is 0 and every output is ln(FLT_EPSILON) = -23 ln 2.
  frame "impulse"  x[n] = A delta[n - j].  u = A delta[n-j] - A/N, y[n] = -(1-c) A/N + A delta[n-j] - c A delta[n-j-1]
                   (the n = 0 rule gives the same constant term), so
                   Y(k) = -(1-c) (A/N) Hh(theta) + A h[j] e^{-i theta j} - c A h[j+1] e^{-i theta (j+1)}
  frame "sine"     x[n] = A sin(w n), w = 2 pi / 16 (1 kHz): 25 whole periods, mean 0.  With g = A (1 - c e^{-i w}):
                   y[n] = Im(g e^{i w n}) for n >= 1 and y[0] = 0 (the formula would give c A sin w there), so
                   Y(k) = (g Hh(theta - w) - conj(g) Hh(theta + w)) / (2 i) - c A sin(w) h[0]

  E_m = sum_k W_m(k) |Y(k)|^2, W_m the triangle of mel bin m evaluated at mel(31.25 k), k = 0 .. 255; out = ln max(E_m, eps).
"""
from __future__ import annotations

import cmath
import json
import math
from pathlib import Path

N, P, NB = 400, 512, 80
C_PRE = 0.97
A_WIN = 2.0 * math.pi / (N - 1)
FLT_EPSILON = 2.0 ** -23


def D(phi: float) -> complex:
    z = cmath.exp(-1j * phi)
    if abs(1.0 - z) < 1e-14:
        return complex(N, 0.0)
    return (1.0 - cmath.exp(-1j * N * phi)) / (1.0 - z)


def Hh(phi: float) -> complex:
    return 0.54 * D(phi) - 0.23 * (D(phi - A_WIN) + D(phi + A_WIN))


def h(n: int) -> float:
    return 0.54 - 0.46 * math.cos(A_WIN * n)


def mel(f: float) -> float:
    return 1127.0 * math.log(1.0 + f / 700.0)


def mel_weight(m: int, k: int) -> float:
    lo, hi = mel(20.0), mel(8000.0)
    d = (hi - lo) / (NB + 1)
    left, center, right = lo + m * d, lo + (m + 1) * d, lo + (m + 2) * d
    x = mel(16000.0 / P * k)
    return max(0.0, min((x - left) / (center - left), (right - x) / (right - center)))


def log_mel(Y) -> list:
    p = [abs(Y(k)) ** 2 for k in range(P // 2)]
    out = []
    for m in range(NB):
        e = sum(mel_weight(m, k) * p[k] for k in range(P // 2))
        out.append(math.log(max(e, FLT_EPSILON)))
    return out


IMPULSE_A, IMPULSE_J = 10000.0, 200
SINE_A, SINE_W = 8000.0, 2.0 * math.pi / 16.0
DC_C = 1000.0


def frames():
    """the three input frames (int16-range floats)"""
    return {"dc": [DC_C] * N,
            "impulse": [IMPULSE_A if n == IMPULSE_J else 0.0 for n in range(N)],
            "sine": [SINE_A * math.sin(SINE_W * n) for n in range(N)]}


def known_answers():
    def y_impulse(k):
        th = 2.0 * math.pi * k / P
        return (-(1.0 - C_PRE) * IMPULSE_A / N * Hh(th) + IMPULSE_A * h(IMPULSE_J) * cmath.exp(-1j * th * IMPULSE_J)
                - C_PRE * IMPULSE_A * h(IMPULSE_J + 1) * cmath.exp(-1j * th * (IMPULSE_J + 1)))

    g = SINE_A * (1.0 - C_PRE * cmath.exp(-1j * SINE_W))

    def y_sine(k):
        th = 2.0 * math.pi * k / P
        return (g * Hh(th - SINE_W) - g.conjugate() * Hh(th + SINE_W)) / 2j - C_PRE * SINE_A * math.sin(SINE_W) * h(0)
    return {"dc": [math.log(FLT_EPSILON)] * NB, "impulse": log_mel(y_impulse), "sine": log_mel(y_sine)}


if __name__ == "__main__":
    out = Path(__file__).resolve().parents[1] / "tests" / "golden" / "fbank_known_answers.json"
    out.write_text(json.dumps({"frame_length": N, "fft": P, "num_mel_bins": NB,
                               "inputs": {"dc": {"C": DC_C}, "impulse": {"A": IMPULSE_A, "j": IMPULSE_J},
                                          "sine": {"A": SINE_A, "w": SINE_W}},
                               "log_mel": known_answers()}, indent=1))
    print(out)
