"""CPU: the device symbol a launcher notes for the profiler (esr_note_kernel) must be the one rocprofv3 prints -- EVERY template argument,
defaulted ones included -- because tools/pmc_traffic.py and bench.py's roofline leg join the two by name.  Round 6: conv64m_kernel gained two
template parameters, RFDN's launchers kept noting three, the join dropped the model's three dominant launches and the bench line's `traffic`
went null.  This test counts: template parameters of each __global__ kernel in csrc/ against the arguments of every name noted for it."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ntire2022_esr_amd", "csrc")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _sources(strip=False):
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".inc", ".h")):
            text = open(os.path.join(CSRC, f)).read()
            if strip:      # comments and preprocessor lines may stand between `template <...>` and `__global__`
                text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
                text = re.sub(r"//[^\n]*", " ", text)
                text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)
            yield f, text


def test_noted_kernel_names_carry_every_template_argument():
    kernels = {}       # name -> number of template parameters (0: not a template)
    tmpl = re.compile(r"(?:template\s*<((?:[^<>]|<[^<>]*>)*)>\s*)?__global__\s+(?:__launch_bounds__\s*\((?:[^()]|\([^()]*\))*\)\s*)?void\s+(\w+)\s*\(")
    for _, text in _sources(strip=True):
        for m in tmpl.finditer(text):
            kernels[m.group(2)] = len(_split_top(m.group(1))) if m.group(1) else 0
    assert len(kernels) >= 20, sorted(kernels)
    noted = []
    for f, text in _sources():
        for m in re.finditer(r'esr_note_kernel\("([^"]*)"', text):
            noted.append((f, m.group(1)))
    assert len(noted) >= 25, noted
    seen = set()
    for f, fmt in noted:
        for part in fmt.split(" + "):
            m = re.fullmatch(r"(\w+)(?:<(.*)>)?", part)
            assert m, (f, fmt)
            name, args = m.group(1), m.group(2)
            assert name in kernels, (f, fmt, "no such __global__ kernel in csrc/")
            n = len(_split_top(args)) if args is not None else 0
            assert n == kernels[name], (f, fmt, f"{name} has {kernels[name]} template parameters, the noted name carries {n}")
            seen.add(name)
    # every kernel of the product path that is launched through esr_run_ops is noted somewhere (research / probe kernels are not)
    for must in ("conv64m_kernel", "rfdb_tail_kernel", "rlfb_chain_kernel", "wino8_f32_kernel", "wino_f32_kernel", "imdb_tail_kernel",
                 "conv_s16_kernel", "conv_f32_kernel", "esa_apply_mfma_kernel", "pack_input_kernel"):
        assert must in seen, must
