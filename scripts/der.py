"""python scripts/der.py ref.rttm hyp.rttm [uri]  — collar 0, overlap scored, optimal mapping (diarizen_amd/der.py)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diarizen_amd.der import der_rttm  # noqa: E402

r = der_rttm(open(sys.argv[1]).read(), open(sys.argv[2]).read(), sys.argv[3] if len(sys.argv) > 3 else None)
print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}))
