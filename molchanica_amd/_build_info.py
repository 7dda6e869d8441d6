"""Revision of the pair kernel's CODE: what a cached measurement of that kernel (profiles/nb_traffic.json: PMC counters cannot be
collected inside bench.py) is keyed by.  Only text that reaches the compiler counts: the two headers that hold the kernel
(mdx_nonbonded_impl.h, mdx_pair_dev.h - with NbArgs) and the parameter structs of mdx_internal.h it reads, each with comments
removed and white space collapsed - so that a comment, or an edit anywhere else in mdx_internal.h, no longer voids the figure
(round 5: a skin-walk commit touching mdx_internal.h nulled `roofline.traffic` in the driver's line)."""
import hashlib
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def _code(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def _struct(text: str, name: str) -> str:
    m = re.search(r"\bstruct\s+" + name + r"\s*\{", text)
    if not m:
        return ""
    depth, i = 0, m.end() - 1
    while i < len(text):
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return text[m.start():i + 1]
        i += 1
    return ""


def pair_kernel_rev() -> str:
    h = hashlib.sha1()
    for f in ("mdx_nonbonded_impl.h", "mdx_pair_dev.h"):
        h.update(_code(open(os.path.join(_CSRC, f)).read()).encode())
    internal = _code(open(os.path.join(_CSRC, "mdx_internal.h")).read())
    for s in ("NbParams", "BondedParams", "WptRule"):
        h.update(_struct(internal, s).encode())
    m = re.search(r"#define MDX_NB_WAVES \d+", internal)
    h.update((m.group(0) if m else "").encode())
    return h.hexdigest()[:12]
