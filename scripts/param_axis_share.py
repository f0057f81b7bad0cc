#!/usr/bin/env python3
"""How much of a model's per-point stage is independent of the parameter(s) that vary across a batch?

BASELINE configs[2] sweeps D5 over 32 parameter rows that differ in ONE parameter (a1 = args[6]); every (parameter row,
tile) workgroup evaluates the whole point stage.  A tile-kernel variant that loops the parameter rows inside the row loop
could evaluate the point-stage values that do not depend on a1 once per grid point instead of 32 times -- IF there are
enough of them.  This script counts them on the generated stage header (the code the kernels really run; CPU only):

  * every statement of the four stages is marked `varying` when it reads args[k], k in the varying set, directly or through
    any stage value it uses;
  * a point-stage statement that is not varying could move out of the parameter loop whole (exact: same operations, same
    operands);
  * of a varying point-stage statement that is a sum, the LEADING terms that are not varying could move too (exact: C adds
    left to right, so that partial sum is the value the statement computes on the way; staging.py `hoist_prefix`), further
    invariant terms only by re-association (not exact, listed separately);
  * every operation is priced in v_fma_f64 issue slots (profiles/r02_valu_rates.txt: mul / add / fma 1, the quick point
    stage's hoisted quotients 3 (pure) / 4, an IEEE division 13, a square root 8 (quick) / 16).

    python scripts/param_axis_share.py d5 6          # model, indices of the varying parameters
"""

import json
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAME = re.compile(r"\b(?:[urcp]_\d+|[urc]_flag|y_\d+)\b")
ARG = re.compile(r"args\[(\d+)\]")
STMT = re.compile(r"^\s*const double (\w+) = (.*);\s*$")
OUT = re.compile(r"^\s*mv\.(\w+) = (.*);\s*$")


def ipow_cost(n):
    return int(math.floor(math.log2(n))) + bin(n).count("1") - 1


def cost(text):
    """Issue slots of one statement's right-hand side (quick point stage spelling)."""
    c = 0.0
    c += 3 * text.count("INFLX_DIVH_PURE(")
    c += 4 * (text.count("INFLX_DIVH(") + text.count("INFLX_DIVS("))
    c += 6 * text.count("INFLX_RCPN(")
    plain = re.sub(r"INFLX_DIV\w*\(", "(", text)
    c += 13 * plain.count("/")
    c += plain.count("*")
    c += len(re.findall(r"(?<=[\w\)\]]) [+-] ", plain))
    for n in re.findall(r"inflx_ipow<(\d+)>", text):
        c += ipow_cost(int(n))
    for n in re.findall(r"(?:inflx_hpow<|INFLX_HPOW\()(\d+)", text):
        c += 8 + ipow_cost(max(1, int(n) // 2)) + 1
    c += 8 * (text.count("INFLX_SQRT(") + len(re.findall(r"(?<![A-Za-z0-9_])sqrt\(", text)))
    c += 70 * len(re.findall(r"(?<![A-Za-z0-9_])(?:sin|cos|tan)\(", text))
    c += 60 * len(re.findall(r"(?<![A-Za-z0-9_])(?:log|exp|sinh|cosh|tanh)\(", text))
    c += 200 * len(re.findall(r"(?<![A-Za-z0-9_])pow\(", text))
    return c


def top_level_terms(text):
    """Split a sum at parenthesis depth 0 into its signed terms, in order."""
    terms, depth, start = [], 0, 0
    i = 0
    while i < len(text):
        ch = text[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        elif depth == 0 and text[i : i + 3] in (" + ", " - ") and i > 0:
            terms.append(text[start:i])
            start = i + 1
            i += 2
        i += 1
    terms.append(text[start:])
    return [t.strip() for t in terms]


def stage_bodies(header):
    """{stage function name: [lines]} of the generated header (the quick point stage when there are two)."""
    bodies, cur = {}, None
    for line in header.splitlines():
        m = re.match(r"INFLX_FN void (inflx_stage_\w+)\(", line)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif line.startswith("}"):
            cur = None
        elif cur:
            bodies[cur].append(line)
    return bodies


def analyse(header, varying):
    bodies = stage_bodies(header)
    point = bodies.get("inflx_stage_point_quick") or bodies["inflx_stage_point"]
    varies = {}

    def depends(text):
        if any(int(k) in varying for k in ARG.findall(text)):
            return True
        return any(varies.get(n, False) for n in NAME.findall(text))

    per_stage = {}
    for fn, tag in (("inflx_stage_uniform", "U"), ("inflx_stage_row", "R"), ("inflx_stage_col", "C")):
        tot = var = n = nv = 0
        for line in bodies[fn]:
            m = STMT.match(line)
            if not m or re.fullmatch(r"[URC]\[\d+\]", m.group(2)):
                continue
            v = depends(m.group(2))
            varies[m.group(1)] = v
            c = cost(m.group(2))
            tot, n = tot + c, n + 1
            if v:
                var, nv = var + c, nv + 1
        per_stage[tag] = {"statements": n, "varying_statements": nv, "slots": tot, "varying_slots": var}
    rec = {"slots": 0.0, "invariant_whole": 0.0, "invariant_leading_terms": 0.0, "invariant_other_terms": 0.0, "statements": []}
    for line in point:
        m = STMT.match(line) or OUT.match(line)
        if not m or re.fullmatch(r"[URC]\[\d+\]", m.group(2)):
            continue
        name, text = m.group(1), m.group(2)
        if name in ("b0", "b1"):  # the basis vector of flag_quantum_dif: dead code in the complete_analysis kernels
            continue
        c = cost(text)
        v = depends(text)
        varies[name] = v
        rec["slots"] += c
        entry = {"name": name, "slots": c, "varying": v}
        if not v:
            rec["invariant_whole"] += c
        else:
            terms = top_level_terms(text)
            if len(terms) > 1:
                flags = [depends(t) for t in terms]
                lead = 0
                while lead < len(terms) and not flags[lead]:
                    lead += 1
                lead_cost = sum(cost(t) for t in terms[:lead]) + max(0, lead - 1) if lead >= 2 else 0.0
                other = sum(cost(t) + 1 for t, f in zip(terms[lead:], flags[lead:]) if not f)
                if lead == 1:
                    other += cost(terms[0])
                rec["invariant_leading_terms"] += lead_cost
                rec["invariant_other_terms"] += other
                entry.update(terms=len(terms), invariant_leading=lead, invariant_elsewhere=int(sum(1 for f in flags[lead:] if not f)))
        rec["statements"].append(entry)
    return per_stage, rec


def main():
    import workloads

    name = sys.argv[1] if len(sys.argv) > 1 else "d5"
    varying = {int(v) for v in sys.argv[2:]} or {6}
    spec, art = workloads.artifact_for(name)
    header = open(art.header_path).read()
    per_stage, rec = analyse(header, varying)
    epilogue = 150.0  # complete_analysis after the model values: ~150 slots per point, all of it downstream of the model values
    exact = rec["invariant_whole"] + rec["invariant_leading_terms"]
    out = {
        "model": name,
        "varying_parameters": {int(k): spec.arg_names[k] for k in sorted(varying)},
        "earlier_stages": per_stage,
        "point_stage_slots": rec["slots"],
        "point_stage_invariant_exact": exact,
        "point_stage_invariant_with_reassociation": exact + rec["invariant_other_terms"],
        "share_of_point_stage_exact": exact / rec["slots"],
        "share_of_point_stage_with_reassociation": (exact + rec["invariant_other_terms"]) / rec["slots"],
        "share_of_point_stage_plus_epilogue_exact": exact / (rec["slots"] + epilogue),
        "statements": rec["statements"],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
