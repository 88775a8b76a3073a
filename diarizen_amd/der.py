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


def parse_uem(text: str) -> Dict[str, List[Tuple[float, float]]]:
    """NIST UEM: `<uri> <channel> <start> <end>` per line -> {uri: [(start, end), ...]} (the scored regions)"""
    out: Dict[str, List[Tuple[float, float]]] = defaultdict(list)
    for line in text.splitlines():
        f = line.split()
        if len(f) >= 4 and not f[0].startswith(";"):
            out[f[0]].append((float(f[2]), float(f[3])))
    return dict(out)


def rttm_uris(text: str) -> List[str]:
    seen = []
    for line in text.splitlines():
        f = line.split()
        if len(f) >= 8 and f[0] == "SPEAKER" and f[1] not in seen:
            seen.append(f[1])
    return seen


def der(ref: List[Turn], hyp: List[Turn], uem: List[Tuple[float, float]] = None, collar: float = 0.0) -> Dict[str, float]:
    """md-eval / dscore semantics (`score.py --collar C`, overlaps scored): time outside the UEM regions is not scored
    (default: everything — dscore then uses the extent of reference and system turns), nor is +-collar around every
    REFERENCE turn boundary; ONE optimal speaker mapping on the scored time."""
    bset = {t for s, e, _ in ref + hyp for t in (s, e)}
    if uem:
        bset |= {t for s, e in uem for t in (s, e)}
    if collar > 0:
        bset |= {t + d for s, e, _ in ref for t in (s, e) for d in (-collar, collar)}
    bounds = sorted(bset)
    if len(bounds) < 2:
        return {"der": 0.0, "miss": 0.0, "false_alarm": 0.0, "confusion": 0.0, "total": 0.0, "mapping": {}}
    mids = 0.5 * (np.array(bounds[:-1]) + np.array(bounds[1:]))
    durs = np.diff(np.array(bounds))
    scored = np.ones(len(mids), dtype=bool)
    if uem:
        scored[:] = False
        for s, e in uem:
            scored |= (mids > s) & (mids < e)
    if collar > 0:
        for s, e, _ in ref:
            for t in (s, e):
                scored &= ~((mids > t - collar) & (mids < t + collar))
    durs = durs * scored

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


def der_rttm(ref_text: str, hyp_text: str, uri: str = None, uem: List[Tuple[float, float]] = None,
             collar: float = 0.0) -> Dict[str, float]:
    return der(parse_rttm(ref_text, uri), parse_rttm(hyp_text, uri), uem=uem, collar=collar)


def score_set(ref_text: str, hyp_texts: Dict[str, str], uem_text: str = None, collar: float = 0.0) -> Dict:
    """A test set the way the recipe scores it (recipes/diar_ssl/run_stage.sh:84-91: dscore `score.py -r <set>/rttm
    -s <out>/*.rttm --collar 0`): one reference RTTM holding every recording, one system RTTM per recording, optional UEM.
    -> {"files": {uri: per-file result}, "overall": errors summed over the files / summed reference time}."""
    uem = parse_uem(uem_text) if uem_text else {}
    files = {}
    for uri, hyp in hyp_texts.items():
        files[uri] = der_rttm(ref_text, hyp, uri=uri, uem=uem.get(uri), collar=collar)
    tot = {k: sum(f[k] for f in files.values()) for k in ("miss", "false_alarm", "confusion", "total")}
    tot["der"] = (tot["miss"] + tot["false_alarm"] + tot["confusion"]) / tot["total"] if tot["total"] > 0 else 0.0
    return {"files": files, "overall": tot, "collar": collar, "missing_in_reference": [u for u in hyp_texts if u not in rttm_uris(ref_text)]}
