"""assert_allclose that also RECORDS how much of its tolerance each call site used.

Every floating-point parity assertion of the GPU tests goes through `close()`; at the end of a session
`tests/conftest.py` writes `gpurun_out/tolerances_observed.json` (one entry per call site: the tolerance in force, the
largest |got - want|, the largest relative error, and `used` = the largest |got - want| / (atol + rtol |want|)).
`tools/tolerance_table.py` turns that file into the table of DESIGN.md section 2; a tolerance is kept at <= 10x what was
observed (SURVEY.md 8(c) contracts: x', r' 1e-12; logw / log-likelihood / ESS 1e-11; estimates 1e-10).
"""
import os
import sys

import numpy as np

RECORDS = {}


def _site():
    f = sys._getframe(2)
    return f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}"


def close(got, want, rtol=1e-7, atol=0.0, err_msg="", what=None):
    g, w = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    try:
        g, w = np.broadcast_arrays(g, w)
        with np.errstate(all="ignore"):
            both = np.isfinite(g) & np.isfinite(w)
            d = np.abs(g - w)[both]
            ref = np.abs(w)[both]
            used = float(np.max(d / (atol + rtol * ref), initial=0.0)) if (atol > 0 or rtol > 0) else 0.0
            rel = float(np.max(d[ref > 0] / ref[ref > 0], initial=0.0))
            mabs = float(np.max(d, initial=0.0))
        key = _site()
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::")[-1]
        rec = RECORDS.setdefault(key, dict(rtol=rtol, atol=atol, used=0.0, max_abs=0.0, max_rel=0.0, n=0, what=what or "",
                                           worst_test=""))
        rec["n"] += int(d.size)
        if used >= rec["used"]:
            rec["used"], rec["worst_test"] = used, test
        rec["max_abs"], rec["max_rel"] = max(rec["max_abs"], mabs), max(rec["max_rel"], rel)
    except Exception:       # recording must never hide the assertion below
        pass
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=err_msg)


def dump(path):
    import json
    if not RECORDS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = {}
    if os.path.exists(path) and os.environ.get("SMCN_TOL_MERGE", "0") == "1":
        old = json.load(open(path))
    old.update(RECORDS)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)
