"""The gfx950 packed-fp32 op_sel erratum behind round 3's overlapped-forward defect (LAB_NOTES.md), reproduced in isolation:
tools/dbg/pk_opsel_probe.hip loops one packed multiply form per victim kernel while a partner kernel on another stream issues MFMAs."""
import os
import re
import subprocess
import warnings

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(REPO, "tools", "dbg", "pk_opsel_probe")


def test_packed_fp32_forms_the_library_uses_are_safe_beside_mfma():
    """Forms 1 (op_sel_hi: a HIGH result half reading a low dword -- how hipcc broadcasts a scalar that has a register of its own, the
    only op_sel use tools/lint_isa.py allows) and 2 (no op_sel) never return a wrong value, alone or beside bf16 / fp32 MFMA partners;
    form 0 (op_sel:[0,1], what the round-3 apply loop compiled to) is reported: on the boxes of round 4 it returns 0 in lanes 48..63
    about 7 times per million wave-instructions beside a bf16 MFMA partner and never alone."""
    if not os.path.exists(PROBE):
        pytest.skip("tools/dbg/pk_opsel_probe not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([PROBE, "0.3", "quick"], capture_output=True, text=True, timeout=300).stdout
    rows = re.findall(r"FORM (\d) .*?\| partner (.*?)\s*: \d+ wave-launches, bad lo (\d+) hi (\d+) of (\S+) wave-instr; lanes \S+ (\d+) (\d+) (\d+) (\d+)", out)
    assert len(rows) == 7 * 3, out
    print(out)
    seen = {}
    for form, partner, lo, hi, n, *lanes in rows:
        seen[(int(form), partner)] = (int(lo) + int(hi), [int(v) for v in lanes])
    for form in (1, 2):
        for partner in ("none", "mfma 16x16x32 bf16", "mfma 16x16x4 f32"):
            assert seen[(form, partner)][0] == 0, (form, partner, seen[(form, partner)])
    for form in range(7):
        assert seen[(form, "none")][0] == 0, (form, seen[(form, "none")])            # no form fails without a partner kernel
    bad, lanes = seen[(0, "mfma 16x16x32 bf16")]
    if bad == 0:
        warnings.warn("pk_opsel_probe: v_pk_mul_f32 op_sel:[0,1] beside an MFMA partner returned no wrong value on this box -- the erratum "
                      "tools/lint_isa.py guards against did not show (driver / firmware change?)")
    else:
        assert lanes[0] == lanes[1] == lanes[2] == 0 and lanes[3] > 0, lanes        # lanes 48..63 only
