"""Diarization error rate (DER) between two RTTMs — the acceptance metric of the reduced-precision modes
(BASELINE.json configs[4]: "DER vs reference on AMI-SDM"; SURVEY §8d: |dDER| <= 0.1 abs for fp16/bf16).

The reference scores with `dscore` (an empty git submodule here, .gitmodules:1-3) / pyannote.metrics (not installed), so
the NIST md-eval definition is implemented directly: collar 0, overlapped speech scored,
    DER = (missed + false alarm + confusion) / total reference speaker time,
with ONE optimal one-to-one speaker mapping (Hungarian on the pairwise overlap durations).  Exact interval
arithmetic on the segment boundaries — no frame grid.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment

Turn = Tuple[float, float, str]     # start, end, speaker


def parse_rttm(text: str, uri: str = None) -> List[Turn]:
    turns = []
    for line in text.splitlines():
        f = line.split()
        if len(f) < 8 or f[0] != "SPEAKER" or (uri is not None and f[1] != uri):
            continue
        start, dur = float(f[3]), float(f[4])
        if dur > 0:
            turns.append((start, start + dur, f[7]))
    return turns


def der(ref: List[Turn], hyp: List[Turn]) -> Dict[str, float]:
    bounds = sorted({t for s, e, _ in ref + hyp for t in (s, e)})
    if len(bounds) < 2:
        return {"der": 0.0, "miss": 0.0, "false_alarm": 0.0, "confusion": 0.0, "total": 0.0, "mapping": {}}
    mids = 0.5 * (np.array(bounds[:-1]) + np.array(bounds[1:]))
    durs = np.diff(np.array(bounds))

    def active(turns):
        spk = sorted({s for _, _, s in turns})
        idx = {s: i for i, s in enumerate(spk)}
        a = np.zeros((len(mids), len(spk)), dtype=bool)
        for s, e, k in turns:
            a[(mids > s) & (mids < e), idx[k]] = True
        return spk, a

    rs, ra = active(ref)
    hs, ha = active(hyp)
    overlap = (((ra[:, :, None] & ha[:, None, :]) * durs[:, None, None]).sum(0) if rs and hs
               else np.zeros((len(rs), len(hs))))
    mapping = {}
    if rs and hs:
        ri, hi = linear_sum_assignment(overlap, maximize=True)
        mapping = {rs[r]: hs[h] for r, h in zip(ri, hi) if overlap[r, h] > 0}
    nr, nh = ra.sum(1), ha.sum(1)
    correct = np.zeros(len(mids))
    for r, h in mapping.items():
        correct += ra[:, rs.index(r)] & ha[:, hs.index(h)]
    total = float((nr * durs).sum())
    miss = float((np.maximum(nr - nh, 0) * durs).sum())
    fa = float((np.maximum(nh - nr, 0) * durs).sum())
    conf = float(((np.minimum(nr, nh) - correct) * durs).sum())
    out = {"miss": miss, "false_alarm": fa, "confusion": conf, "total": total, "mapping": mapping}
    out["der"] = (miss + fa + conf) / total if total > 0 else (0.0 if fa == 0 else float("inf"))
    return out


def der_rttm(ref_text: str, hyp_text: str, uri: str = None) -> Dict[str, float]:
    return der(parse_rttm(ref_text, uri), parse_rttm(hyp_text, uri))
