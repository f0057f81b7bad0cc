"""Axis staging: turn the five model expressions into four straight-line device functions.

The sweep evaluates V, v00, v10, v11 and |dV|^2 on a grid x[0] = row axis, x[1] = column axis.
Every sub-expression is classified by the axes it depends on

    U  parameters only      evaluated once per thread at kernel start (wave-uniform)
    R  x[0] (+ parameters)  evaluated once per grid row,    kept in LDS
    C  x[1] (+ parameters)  evaluated once per grid column, kept in registers
    P  both                 evaluated per grid point

and emitted into the stage it belongs to (``inflx_stage_uniform/_row/_col/_point``).

Hard requirement: *the arithmetic must stay the reference's arithmetic*.  The reference prints each
expression with sympy's C printer and evaluates it as written; at singular points of a model
(0/0 forms where a basis vector degenerates, e.g. theta = k*pi in the D5 model) the result -- NaN or
a cancellation artefact -- depends on the exact form of the expression, and any algebraic rewrite
(``sympy.cse`` rebuilds expressions with auto-evaluation, which cancels factors) changes it.  So
this module never builds new sympy expressions from rewritten pieces.  It only

  * shares *structurally identical* sub-expressions (the same node is computed once: exact);
  * moves *whole* sub-expressions to the stage of the axes they depend on (exact);
  * moves the *leading* partial product ``((a*b)*c)`` of an n-ary product (the leading partial sum of
    an n-ary sum, a whole denominator ``/(d1*d2*d3)``) to the lower stage when those leading operands
    depend on fewer axes: C evaluates left to right, so that partial result is the very value the
    reference computes on the way (exact; ``test_staging_does_not_change_a_single_bit``);
  * replaces ``pow(x, n)`` for small integer / half-integer n by a multiplication chain (the one
    deliberate rounding-level difference of the default mode).

Re-association of products and sums by axis class (``group_quotient`` / ``group_terms``) exists only in
the opt-in fast mode ``Compiler(regroup=True)``; it is not exact and never part of the parity path.

Everything is done while printing: :class:`HIPInflatoxPrinter` asks the :class:`Stager` for the
C expression of every child node, and the stager answers with either the child's own printed form
or the name of a stage variable that holds it.
"""

from __future__ import annotations

import math
import re
import sys
from collections import Counter, defaultdict

import sympy
from sympy.core.mul import _keep_coeff
from sympy.printing.c import C99CodePrinter
from sympy.printing.precedence import PRECEDENCE, precedence

# dependence masks: bit 0 = depends on x[0] (row axis), bit 1 = depends on x[1] (column axis)
U, R, C, P = 0, 1, 2, 3
STAGE_PREFIX = {U: "u", R: "r", C: "c", P: "p"}
ARRAY = {U: "U", R: "R", C: "C"}
_STAGE_NAME = re.compile(r"\b[urcp]_\d+\b")
_SQRT_CALL = re.compile(r"(?<![A-Za-z0-9_])sqrt\(")
_HPOW_CALL = re.compile(r"(?<![A-Za-z0-9_])inflx_hpow<(\d+)>\(")
OUTPUT_FIELDS = ("V", "v00", "v10", "v11", "g", "b0", "b1")


def split_product(expr):
    """The reference printer's view of a product (sympy CodePrinter._print_Mul): a sign, the
    numerator factors and the denominator factors (as positive powers), in printed order."""
    c, e = expr.as_coeff_Mul()
    sign = ""
    if c < 0:
        expr = _keep_coeff(-c, e)
        sign = "-"
    num, den, paren = [], [], []
    args = expr.as_ordered_factors() if expr.is_Mul else [expr]
    for item in args:
        if item.is_commutative and item.is_Pow and item.exp.is_Rational and item.exp.is_negative:
            if item.exp != -1:
                den.append(sympy.Pow(item.base, -item.exp, evaluate=False))
            else:
                if len(item.args[0].args) != 1 and isinstance(item.base, sympy.Mul):
                    paren.append(item)
                den.append(sympy.Pow(item.base, -item.exp))
        else:
            num.append(item)
    return sign, (num or [sympy.S.One]), den, paren


def printed_children(e):
    """Nodes the printer will ask for when it prints ``e`` (used for reference counting)."""
    if e.is_Mul:
        _, num, den, _ = split_product(e)
        return num + den
    return list(e.args)


class HIPInflatoxPrinter(C99CodePrinter):
    """C99 printer for device code whose child printing is routed through a :class:`Stager`.

    Symbols resolve through an explicit table (fields -> ``x0``/``x1``, parameters -> ``args[k]``);
    an unknown symbol is an error, because parameter numbering was fixed beforehand by the
    reference-order registration pass (compiler.Compiler._number_parameters).
    """

    #: pow(x, n) becomes a multiplication chain for 2 <= |n| <= 16 (half-integer n/2 for |n| <= 33); the chain's
    #: error grows like n/2 ulp against < 1 ulp for libm's pow, so larger exponents keep the pow() call.  The
    #: example models use nothing beyond x^12.  A stand-alone negative power prints as 1.0/chain: where x^n
    #: overflows this gives 0 while pow(x, -n) would still return a denormal -- accepted (inside products sympy
    #: prints negative powers as denominators, exactly as the reference's C has them).
    MAX_INT_POW = 16

    def __init__(self, names: dict, stager=None):
        super().__init__()
        self.names = names
        self.stager = stager
        # constants print as INFLX_<macro>; the generated header defines them
        self.math_macros = {k: "INFLX_" + v for k, v in self.math_macros.items()}

    # -- routing ---------------------------------------------------------------------------------
    def _print(self, expr, **kwargs):
        if self.stager is not None and isinstance(expr, sympy.Basic):
            return self.stager.value(expr)
        return super()._print(expr, **kwargs)

    def print_node(self, expr):
        """Print ``expr``'s own operator (children go through :meth:`_print`, i.e. the stager)."""
        return super()._print(expr)

    # -- atoms -----------------------------------------------------------------------------------
    def _print_Symbol(self, expr):
        try:
            return self.names[expr]
        except KeyError:
            raise KeyError(f"symbol {expr!r} was not registered before printing") from None

    def _print_Integer(self, expr):
        v = int(expr)
        return str(v) if abs(v) < 2**31 else f"{v}.0"

    # -- operators -------------------------------------------------------------------------------
    def _print_Pow(self, expr):
        base, exp = expr.base, expr.exp
        if exp.is_Integer:
            n = int(exp)
            if 2 <= abs(n) <= self.MAX_INT_POW:
                body = f"inflx_ipow<{abs(n)}>({self._print(base)})"
                return body if n > 0 else f"(1.0/{body})"
        elif exp.is_Rational and exp.q == 2:
            n = int(exp.p)
            if 3 <= abs(n) <= 2 * self.MAX_INT_POW:
                body = f"inflx_hpow<{abs(n)}>({self._print(base)})"
                return body if n > 0 else f"(1.0/{body})"
            if n == -1:
                # a stand-alone x**(-1/2) (inside a product sympy prints 1/sqrt(x) itself; on its own --
                # e.g. as a cse definition -- C99 gets pow(x, -1.0/2.0), compiler.py:403-408): one sqrt and
                # one division instead of the general pow; `+ 0.0` keeps pow's +inf at x = -0.0
                return f"(1.0/sqrt({self._print(base)} + 0.0))"
        return super()._print_Pow(expr)

    # -- special functions (reference: GSLInflatoxPrinter, compiler.py:123-212 -> gsl_sf_bessel_*) ------
    # device counterparts live in csrc/inflx_sf.h: integer orders by name / recurrence, real orders (a non-integer number,
    # a model parameter, any expression) through inflx_sf_bessel_{J,Y,I,K}nu -- the reference's gsl_sf_bessel_*nu
    _CYLINDRICAL = {"besselj": "J", "bessely": "Y", "besseli": "I", "besselk": "K"}
    _SPHERICAL = {"jn": "j", "yn": "y"}

    def _bessel(self, expr, letter, named_orders, general):
        nu, arg = expr.args
        x = self._print(arg)
        if not (nu.is_number and nu.is_integer):
            if general != "n":
                # the reference has no non-integer spherical functions either (compiler.py:196-198: "No non-integer impl found.")
                raise KeyError(f"{expr.func.__name__} of non-integer order {nu}: no non-integer implementation (here as in the reference)")
            return f"inflx_sf_bessel_{letter}nu({self._print(nu)}, {x})"
        n = int(nu)
        if n in named_orders:
            return f"inflx_sf_bessel_{letter}{n}({x})"
        return f"inflx_sf_bessel_{letter}{general}({n}, {x})"

    def _print_besselj(self, expr):
        return self._bessel(expr, "J", (0, 1), "n")

    def _print_bessely(self, expr):
        return self._bessel(expr, "Y", (0, 1), "n")

    def _print_besseli(self, expr):
        return self._bessel(expr, "I", (0, 1), "n")

    def _print_besselk(self, expr):
        return self._bessel(expr, "K", (0, 1), "n")

    def _print_jn(self, expr):
        return self._bessel(expr, "j", (0, 1, 2), "l")

    def _print_yn(self, expr):
        return self._bessel(expr, "y", (0, 1, 2), "l")

    def _print_hyper(self, expr):
        ap, bq, x = expr.args
        kind = (len(ap), len(bq))
        if kind in ((0, 1), (1, 1), (2, 1), (2, 0)):
            operands = ", ".join(self._print(v) for v in list(ap) + list(bq) + [x])
            return f"inflx_sf_hyperg_{kind[0]}F{kind[1]}({operands})"
        # the reference's printer refuses these as well (compiler.py:173-175)
        raise NotImplementedError(f"hypergeometric function {kind[0]}F{kind[1]}: only 0F1, 1F1, 2F1 and 2F0 exist (here as in the reference)")

    def _operand(self, item, level):
        """``parenthesize`` for an operand that may have been replaced by a stage variable."""
        text = self._print(item)
        if self.stager is not None and self.stager.is_named(item):
            return text
        return f"({text})" if precedence(item) <= level else text  # CodePrinter.parenthesize, strict=False

    def _print_Mul(self, expr):
        prec = precedence(expr)
        sign, num, den, paren = split_product(expr)
        if self.stager is not None:
            num, den = self.stager.group_quotient(num, den)
            num = self.stager.hoist_prefix(num, "*")
            den = self.stager.hoist_prefix(den, "*")
        if len(num) == 1 and sign == "-":
            num_s = [self._operand(num[0], 0.5 * (PRECEDENCE["Pow"] + PRECEDENCE["Mul"]))]
        else:
            num_s = [self._operand(x, prec) for x in num]
        den_s = [self._operand(x, prec) for x in den]
        for item in paren:
            if item.base in den:
                k = den.index(item.base)
                if not den_s[k].startswith("("):
                    den_s[k] = f"({den_s[k]})"
        if not den:
            return sign + "*".join(num_s)
        if len(den) == 1:
            recip = self.stager.hoisted_reciprocal(den) if self.stager is not None else None
            if recip is not None:
                if self.stager.hoist_inline:
                    # self-checking quotient: no `ok` flag, no second copy of the point stage (inflx_div_by_hoisted_inline)
                    return f"INFLX_DIVI({sign}{'*'.join(num_s)}, {den_s[0]}, {recip})"
                macro = "INFLX_DIVH_PURE" if self.stager.pure_numerator(num, num_s, den[0], den_s[0]) else "INFLX_DIVH"
                return f"{macro}({sign}{'*'.join(num_s)}, {den_s[0]}, {recip})"
            shared = self.stager.shared_reciprocal(den, den_s[0]) if self.stager is not None else None
            if shared is not None:
                return f"INFLX_DIVS({sign}{'*'.join(num_s)}, {den_s[0]}, {shared})"
            return sign + "*".join(num_s) + "/" + den_s[0]
        # a/(d1*d2*d3): C multiplies the denominator out first and divides once
        den_text = "(" + "*".join(den_s) + ")"
        shared = self.stager.shared_reciprocal(den, den_text) if self.stager is not None else None
        if shared is not None:
            return f"INFLX_DIVS({sign}{'*'.join(num_s)}, {den_text}, {shared})"
        return sign + "*".join(num_s) + "/" + den_text

    def _print_Add(self, expr, order=None):
        terms = self._as_ordered_terms(expr, order=order)
        if self.stager is not None:
            terms = self.stager.hoist_prefix(self.stager.group_terms(terms), "+")
        prec = precedence(expr)
        parts = []
        for k, term in enumerate(terms):
            if self.stager is not None:
                # (clang fuses ONE operand of an addition, the left one first: in ((t0 + t1) + t2) ... the product t1 stays a rounded
                # value when t0 is a product itself)
                t, named = self.stager.sum_operand(term, fusable=not (k == 1 and self.stager.is_plain_product(terms[0])))
            else:
                t, named = self._print(term), False
            if t.startswith("-") and not (term.is_Add and not named):
                sign, t = "-", t[1:]
            else:
                sign = "+"
            if not named and (precedence(term) < prec or term.is_Add):
                t = f"({t})"
            parts.extend([sign, t])
        first = parts.pop(0)
        return ("" if first == "+" else first) + " ".join(parts)

    # functions without a C99 spelling
    def _print_coth(self, e):
        return f"inflx_coth({self._print(e.args[0])})"

    def _print_sech(self, e):
        return f"inflx_sech({self._print(e.args[0])})"

    def _print_csch(self, e):
        return f"inflx_csch({self._print(e.args[0])})"

    def _print_cot(self, e):
        return f"inflx_cot({self._print(e.args[0])})"

    def _print_sec(self, e):
        return f"inflx_sec({self._print(e.args[0])})"

    def _print_csc(self, e):
        return f"inflx_csc({self._print(e.args[0])})"


class _Group(sympy.AtomicExpr):
    """Placeholder (fast mode) for 'the operands of one axis class inside an n-ary product or sum':
    op "/" = product of ``items`` divided by the product of ``den``; op "+" = sum of ``items``."""

    is_commutative = True
    is_number = False

    def __new__(cls, op, items, den=()):
        obj = sympy.AtomicExpr.__new__(cls)
        obj.op = op
        obj.items = tuple(items)
        obj.den = tuple(den)  # op "/": items / den
        return obj

    def _hashable_content(self):
        return (self.op, self.items, self.den)

    @property
    def free_symbols(self):
        return set().union(*[i.free_symbols for i in self.items + self.den])


class Stager:
    """Decides, node by node, where a sub-expression is evaluated, and collects the stage code.

    ``staged=False`` turns both sharing and hoisting off: the five expressions are printed as the
    reference would print them (modulo the pow chains) and evaluated per grid point -- the parity
    switch that mirrors the reference's five separate C functions.
    """

    def __init__(self, functions, x0, x1, names, staged=True, regroup=False, hoist_reciprocals=False, shared_point_dens=(), contract_products=False):
        """``functions``: one ``(replacements, [expressions])`` pair per generated C function of the
        reference (five scalar functions and the two-component basis vector ``v``), where
        ``replacements`` is the (possibly empty) list of ``(symbol, definition)`` pairs the
        reference's per-function ``sympy.cse`` produced; those symbols are local to their function."""
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 50000))
        self.regroup = regroup
        self.staged = staged
        # a product that is a direct operand of a sum keeps its LAST multiplication in the statement of the sum (see sum_operand)
        self.contract_products = bool(contract_products) and staged
        # "inline": the hoisted quotients check themselves and fall back to the IEEE division on the spot (one point stage,
        # no `ok` bookkeeping); True: the quick / IEEE pair of point stages with the row redone when a quotient is irregular
        self.hoist_inline = hoist_reciprocals == "inline" and staged
        self.hoist_reciprocals = bool(hoist_reciprocals) and staged
        self.x0, self.x1 = x0, x1
        self.printer = HIPInflatoxPrinter(names, self)
        self.lines = {U: [], R: [], C: [], P: []}
        self.stage_of = {}  # name -> stage
        self.named = {}  # node -> name
        self.local = {}  # function-local cse symbol -> (name, mask)
        self.used_by = defaultdict(set)  # name -> stages (or "out") that read it
        self._mask = {}
        self._count = defaultdict(int)
        self._by_text = {}  # (stage, right-hand side text) -> name of the variable that holds it
        self.range_terms = {U: {}, R: {}, C: {}}  # operand text -> E: |operand| must lie in [2^-E, 2^E] (pure hoisted quotients)
        self.point_den_counts = Counter()  # text of a per-point denominator -> quotients of the point stage that divide by it
        self.shared_point_dens = frozenset(shared_point_dens) if self.hoist_reciprocals else frozenset()
        self._shared_recip = {}  # denominator text -> point-stage variable holding its refined reciprocal
        self._ctx = P
        self.refs = Counter()
        # identical nodes may be shared across the five functions only if no function-local
        # symbols exist (a node mentioning `cse3` means something else in every function)
        share_across = all(not repl for repl, _ in functions)
        if share_across:
            self._count_refs([e for _, exprs in functions for e in exprs])
        # `regroup` may name the functions (positions in `functions`: 0 V, 1 v00, 2 v10, 3 v11, 4 |dV|^2, 5 the basis
        # vector) whose products and sums may be re-associated; the others keep the reference's arithmetic bit for bit.
        # Those are printed FIRST: a sub-expression they share with a regrouped function then exists in its exact form and
        # the regrouped function reuses it, never the other way round.
        regroup_set = None if isinstance(regroup, bool) else frozenset(regroup)
        order = list(range(len(functions)))
        if regroup_set is not None:
            order.sort(key=lambda k: k in regroup_set)
        slots = {}
        for k in order:
            repl, exprs = functions[k]
            self.regroup = (k in regroup_set) if regroup_set is not None else bool(regroup)
            self.outputs, self.out_masks = [], []
            self._print_function(repl, exprs, share_across, staged)
            slots[k] = (self.outputs, self.out_masks)
        self.regroup = bool(regroup_set) if regroup_set is not None else bool(regroup)
        self.outputs = [t for k in range(len(functions)) for t in slots[k][0]]
        self.out_masks = [m for k in range(len(functions)) for m in slots[k][1]]
        # the axis mask that selects the row-broadcast kernels covers the five values the sweeps use
        self.out_mask = 0
        for m in self.out_masks[:5]:
            self.out_mask |= m
        self.exports = {m: [n for n, s in self.stage_of.items() if s == m and (self.used_by[n] - {m})] for m in (U, R, C)}

    def _print_function(self, repl, exprs, share_across, staged):
        if True:
            if not share_across:
                self.named, self._mask, self.refs, self.local = {}, {}, Counter(), {}
                self._count_refs([d for _, d in repl] + list(exprs))
            for sym, definition in repl:
                m = self.mask(definition) if staged else P
                self._ctx = m
                self.local[sym] = (self._variable(definition, m, reference=False), m)
            for expr in exprs:
                self._ctx = P
                self.out_masks.append(self.mask(expr) if staged else P)
                text = self.value(expr)
                if text in self.stage_of:
                    self.used_by[text].add("out")
                self.outputs.append(text)

    def _count_refs(self, roots):
        if not self.staged:
            return
        seen = set()
        stack = list(roots)
        while stack:
            e = stack.pop()
            if e.is_Atom:
                continue
            self.refs[e] += 1
            if e in seen:
                continue
            seen.add(e)
            stack.extend(printed_children(e))

    # -- classification --------------------------------------------------------------------------
    def mask(self, e):
        if e == self.x0:
            return R
        if e == self.x1:
            return C
        if isinstance(e, _Group):
            m = 0
            for i in e.items + e.den:
                m |= self.mask(i)
            return m
        if e.is_Atom:
            return self.local[e][1] if e in self.local else U
        got = self._mask.get(e)
        if got is None:
            got = 0
            for a in e.args:
                got |= self.mask(a)
                if got == P:
                    break
            self._mask[e] = got
        return got

    def is_named(self, e):
        return e in self.named

    # -- the two questions the printer asks ----------------------------------------------------------
    def value(self, e):
        """C text for node ``e`` as an operand in the stage currently being printed."""
        if isinstance(e, _Group):
            return self._variable(e, self.mask(e))
        if e.is_Atom:
            return self._reference(self.local[e][0]) if e in self.local else self.printer.print_node(e)
        if not self.staged:
            return self.printer.print_node(e)
        if e in self.named:
            return self._reference(self.named[e])
        if not e.free_symbols:
            return self.printer.print_node(e)  # pure number: the device compiler folds it
        if e.is_Mul and len(e.args) == 2 and e.args[0] is sympy.S.NegativeOne and e.args[1].is_Atom:
            return self.printer.print_node(e)  # -symbol: cheaper to negate in place than to stage
        m = self.mask(e)
        if m != self._ctx or self.refs[e] >= 2:
            return self._variable(e, m)
        return self.printer.print_node(e)

    @staticmethod
    def is_plain_product(term) -> bool:
        """Does the C text of ``term`` end in a multiplication (a*b, -a*b*c, 2*x; not a*b/c, not a function call)?"""
        if isinstance(term, _Group) or not term.is_Mul:
            return False
        _, num, den, _ = split_product(term)
        return not den and len(num) >= 2

    def sum_operand(self, term, fusable=True):
        """(C text, is it a stage variable?) of one operand of a sum.

        ``contract_products`` (``Compiler(contraction="expression")``): the reference's C holds every model function as ONE expression,
        and the compiler it is built with -- ``zig cc`` = clang, ``-ffp-contract=on`` -- turns ``x*y + z`` into fma(x, y, z) wherever a
        multiplication is a direct operand of an addition in that expression (python/inflatox/compiler.py:299-310,575-584).  hipcc is
        the same clang with the same rule, but the rule ends at a statement: a product that the stager has made a variable of -- shared
        by several uses, or evaluated once per row -- reaches the sum already rounded.  Here such a product is spelled out in the sum's
        own statement instead, its leading factors as the (stage) value the reference computes on the way and its LAST multiplication in
        place, so that the compiler fuses exactly what clang fuses in the reference's text: ``r_12 + p_4`` becomes ``r_1*r_2 + p_4``.
        Same instruction count per point (an fma for an add), a few more row values read."""
        if self.contract_products and fusable and not isinstance(term, _Group) and term.is_Mul and term not in self.local:
            sign, num, den, _ = split_product(term)
            m = self.mask(term)
            # only a product that would arrive as a variable: one of an earlier stage, or one shared by several uses
            would_be_named = term.free_symbols and (m != self._ctx or self.refs[term] >= 2 or term in self.named)
            if would_be_named and not den and len(num) >= 2:
                if m == self._ctx:
                    return self.printer.print_node(term), False  # (its leading factors of earlier stages still become one stage value)
                prec = precedence(term)
                head, last = num[:-1], num[-1]
                head_text = self.value(_Group("*prefix", head)) if len(head) >= 2 else self.printer._operand(head[0], prec)
                return f"{sign}{head_text}*{self.printer._operand(last, prec)}", False
        return self.value(term), self.is_named(term)

    def group_terms(self, items):
        """Fast mode only (``regroup``): within a sum printed in stage ctx, the terms of each lower class
        (>= 2 of them) are added first and become one stage variable; printed order is kept otherwise."""
        op = "+"
        if not self.staged or len(items) < 2 or not self.regroup:
            return items
        ctx = self._ctx
        by = defaultdict(list)
        for it in items:
            m = self.mask(it)
            if m != ctx and it.free_symbols:
                by[m].append(it)
        merged, done = [], set()
        for it in items:
            m = self.mask(it)
            if m != ctx and it.free_symbols and len(by[m]) >= 2:
                if m not in done:
                    done.add(m)
                    merged.append(_Group(op, by[m]))
            else:
                merged.append(it)
        return merged

    def hoist_prefix(self, items, op):
        """EXACT: C evaluates ``a*b*c*d`` as ``((a*b)*c)*d`` and ``a+b+c`` as ``(a+b)+c``.  If the first
        k >= 2 operands of a product (or sum) printed in stage ctx together depend on fewer axes than
        ctx, their partial product (sum) is the very value the reference computes on the way, so it can
        be evaluated once in the lower stage without changing a bit.  (A denominator ``/(d1*d2*d3)`` is a
        product of its own, so a row-only denominator becomes one row-stage value.)"""
        if not self.staged or len(items) < 2:
            return items
        ctx = self._ctx
        cumulative, k, best = 0, 0, None
        for it in items:
            cumulative |= self.mask(it)
            if cumulative == ctx:
                break
            k += 1
            if k >= 2 and any(i.free_symbols for i in items[:k]):
                best = (k, cumulative)
        if best is None:
            return items
        k, m = best
        if k == len(items) and m == ctx:
            return items
        return [_Group(op + "prefix", items[:k])] + list(items[k:])

    def hoisted_reciprocal(self, den):
        """A quotient printed in the per-point stage whose whole denominator is known one stage earlier is
        evaluated as ``inflx_div_by_hoisted(num, den, y)`` with ``y = inflx_recip(den)`` a value of that
        earlier stage (csrc/inflx_device_math.h explains why this is the correctly rounded quotient).
        Returns the text of ``y`` or None when the quotient does not qualify."""
        if not self.hoist_reciprocals or self._ctx != P or len(den) != 1:
            return None
        d = den[0]
        if not d.free_symbols or self.mask(d) == P:
            return None
        return self._variable(_Group("recip", [d]), self.mask(d))

    def shared_reciprocal(self, den, den_text):  # den: the factors of the denominator
        """A quotient of the point stage by a PER-POINT denominator: counted, and -- when the same denominator (same
        text, hence same value) serves several quotients and the first pass has said so -- divided through the shared
        refined reciprocal ``INFLX_RCPN(den)`` (csrc/inflx_device_math.h: inflx_shared_reciprocal / inflx_div_by_shared,
        the compiler's own division sequence with its reciprocal computed once).  Returns the variable's name or None."""
        mask = 0
        for d in den:
            mask |= self.mask(d)
        if not self.staged or self._ctx != P or mask != P:
            return None
        self.point_den_counts[den_text] += 1
        if den_text not in self.shared_point_dens:
            return None
        name = self._shared_recip.get(den_text)
        if name is None:
            # (a numbering of its own: the p_k of the second pass must be the p_k of the first, whose texts selected the denominators)
            name = f"y_{len(self._shared_recip)}"
            self.stage_of[name] = P
            self.lines[P].append(f"  const double {name} = INFLX_RCPN({den_text});")
            self._shared_recip[den_text] = name
        return self._reference(name)

    PURE_MAX_FACTORS = 4
    PURE_BUDGET = 480  # |log2| allowed for the whole numerator; a numerator of k factors allows 480/k per factor

    def pure_numerator(self, num, num_s, den, den_s):
        """Is the numerator of this hoisted quotient a product of at most PURE_MAX_FACTORS values of earlier
        stages (and modest numeric constants)?  Then whether Markstein's three operations are valid
        (csrc/inflx_device_math.h) does not depend on the grid point but on the row, the column and the
        parameters separately: with each of k factors in [2^-E, 2^E], E = 480/k, and the denominator in
        [2^-500, 2^500], the numerator lies in [2^-490, 2^490] and nothing can overflow, underflow or lose the
        exactness of the residual.  The factors are recorded per stage; each stage exports ONE flag (0.0 = all
        its operands in range), the point stage tests the three flags once instead of comparing every quotient."""
        if not self.hoist_reciprocals or self._ctx != P:
            return False
        factors = []
        budget = 0.0
        for item, text in zip(num, num_s):
            if not item.free_symbols:
                try:
                    v = abs(float(item))
                except (TypeError, ValueError):
                    return False
                if v == 0.0:
                    return False
                budget += abs(math.log2(v))
                continue
            m = self.mask(item)
            if m == P:
                return False
            factors.append((m, text.lstrip("-")))
        if not factors or len(factors) > self.PURE_MAX_FACTORS or budget > 10.0:
            return False
        e = self.PURE_BUDGET // len(factors)
        for m, text in factors:
            self.range_terms[m][text] = min(e, self.range_terms[m].get(text, e))
        self.range_terms[self.mask(den)][den_s] = min(500, self.range_terms[self.mask(den)].get(den_s, 500))
        return True

    def finish_range_flags(self):
        """One exported flag per stage that has range terms; returns the names the point stage must test."""
        names = []
        for m in (U, R, C):
            if self.range_terms[m]:
                name = f"{STAGE_PREFIX[m]}_flag"
                self.stage_of[name] = m
                terms = [f"inflx_out_of_range<{e}>({text})" for text, e in self.range_terms[m].items()]
                self.lines[m].append(f"  const double {name} = " + " + ".join(terms) + ";")
                self.used_by[name].add(P)
                names.append(name)
        self.exports = {m: [n for n, st in self.stage_of.items() if st == m and (self.used_by[n] - {m})] for m in (U, R, C)}
        return names

    def group_quotient(self, num, den):
        """Fast mode only (``regroup``): inside a product printed in stage ctx, the numerator and
        denominator factors of each lower class become ONE stage variable num/den, so that the division
        is done once per row/column instead of once per point.  Not exact: (a*c)/b becomes c*(a/b)."""
        if not self.staged or not self.regroup:
            return num, den
        ctx = self._ctx

        def movable(it):
            # numeric constants travel with the parameter-only class (a constant denominator such as
            # pi^3 would otherwise keep a per-point division alive)
            return self.mask(it) != ctx and (it.free_symbols or (ctx != U and not it.is_Rational))

        classes = {}
        for it in num:
            if movable(it):
                classes.setdefault(self.mask(it), ([], []))[0].append(it)
        for it in den:
            if movable(it):
                classes.setdefault(self.mask(it), ([], []))[1].append(it)
        # parameter-only factors ride along with the row (else column) factors of the same product:
        # one multiplication less per point and one exported value less
        fold = {}
        if ctx == P and U in classes and (R in classes or C in classes):
            target = R if R in classes else C
            classes[target][0].extend(classes[U][0])
            classes[target][1].extend(classes[U][1])
            del classes[U]
            fold[U] = target
        merged = {m: _Group("/", n, d) for m, (n, d) in classes.items() if (len(n) + len(d) >= 2 or d) and any(i.free_symbols for i in n + d)}
        if not merged:
            return num, den
        for src, dst in fold.items():
            if dst in merged:
                merged[src] = merged[dst]
        new_num, done = [], set()
        for it in num:
            m = self.mask(it)
            if m in merged and movable(it):
                m = fold.get(m, m)
                if m not in done:
                    done.add(m)
                    new_num.append(merged[m])
            else:
                new_num.append(it)
        for m, g in merged.items():
            if fold.get(m, m) not in done:
                done.add(fold.get(m, m))
                new_num.append(g)  # class present in the denominator only: contributes 1/den
        new_den = [it for it in den if not (self.mask(it) in merged and movable(it))]
        if new_num and new_num[0] is sympy.S.One and len(new_num) > 1:
            new_num = new_num[1:]
        return new_num, new_den

    # -- stage variables ---------------------------------------------------------------------------
    def _reference(self, name):
        self.used_by[name].add(self._ctx)
        return name

    def _variable(self, e, m, reference=True):
        name = self.named.get(e)
        if name is None:
            saved = self._ctx
            self._ctx = m
            if isinstance(e, _Group):
                if e.op == "*prefix":
                    text = "*".join(self.printer._operand(i, PRECEDENCE["Mul"]) for i in e.items)
                elif e.op == "+prefix":
                    text = self._sum_text(e.items)
                elif e.op == "recip":
                    text = f"inflx_recip({self.printer._print(e.items[0])})"
                elif e.op == "/":
                    top = "*".join(self.printer._operand(i, PRECEDENCE["Mul"]) for i in e.items) if e.items else "1.0"
                    bottom = "*".join(self.printer._operand(i, PRECEDENCE["Mul"]) for i in e.den)
                    text = top if not e.den else (f"{top}/{bottom}" if len(e.den) == 1 else f"{top}/({bottom})")
                else:
                    text = self._sum_text(e.items)
            else:
                text = self.printer.print_node(e)
            self._ctx = saved
            # Value numbering on the emitted text: a statement whose right-hand side is, character for
            # character, one that was emitted before (operands are literals, args[k], x0/x1 and stage
            # variables, so equal text means equal value) is that earlier variable.  This is what shares work
            # between the five functions when the reference ran sympy.cse per function (cse=True): there every
            # function has cse symbols of its own, nodes cannot be compared across functions, and without
            # this each function re-evaluated the sub-expressions it has in common with the others.  Exact:
            # statements are merged whole, no statement is rewritten.
            name = self._by_text.get((m, text))
            if name is None:
                k = self._count[m]
                self._count[m] += 1
                name = f"{STAGE_PREFIX[m]}_{k}"
                self.stage_of[name] = m
                self.lines[m].append(f"  const double {name} = {text};")
                self._by_text[(m, text)] = name
            self.named[e] = name
        return self._reference(name) if reference else name

    def _sum_text(self, terms):
        parts = []
        for k, term in enumerate(terms):
            t, named = self.sum_operand(term, fusable=not (k == 1 and self.is_plain_product(terms[0])))
            if t.startswith("-") and not (term.is_Add and not named):
                sign, t = "-", t[1:]
            else:
                sign = "+"
            if not named and (precedence(term) < PRECEDENCE["Add"] or term.is_Add):
                t = f"({t})"
            parts.extend([sign, t])
        first = parts.pop(0)
        return ("" if first == "+" else first) + " ".join(parts)

    def statement_counts(self):
        return {m: len(v) for m, v in self.lines.items()}


def emit_stage_header(
    model, param_slots: dict, constants: dict, model_name: str, version: str, abi_version: str, staged: bool = True, cse=None, cse_vector=None, regroup: bool = False,
    hoist_reciprocals: bool = False,
    share_point_reciprocals: bool = False,
    quick_sqrt: bool | None = None,
    contract_products: bool = False,
):
    """Return (header text, info dict) for the model.

    ``cse``: ``None``, or a callable ``expr -> (replacements, reduced)`` reproducing the reference's
    per-function ``sympy.cse`` call (compiler.py:403-410); when given, the reference evaluates the
    cse'd form, so that form -- not the plain expression -- is what gets staged.  ``cse_vector``
    is the list-valued variant the reference uses for the basis vectors (compiler.py:425-433).
    The defaults of ``share_point_reciprocals`` (off) and ``quick_sqrt`` (``None``: follows ``hoist_reciprocals``) are
    ``Compiler``'s, so that a direct caller generates the point stage a user gets."""
    if quick_sqrt is None:
        quick_sqrt = hoist_reciprocals is True or hoist_reciprocals == 1
    x0, x1 = model.coordinates
    exprs = [
        sympy.sympify(model.potential),
        sympy.sympify(model.hesse_cmp[0][0]),
        sympy.sympify(model.hesse_cmp[1][0]),
        sympy.sympify(model.hesse_cmp[1][1]),
        sympy.sympify(model.gradient_square),
    ]
    basis_v = [sympy.sympify(c) for c in model.basis[0]]  # the C function `v` (Potential::grad)
    tangents = set(model.coordinate_tangents)
    for e in exprs + basis_v:
        if e.free_symbols & tangents:
            raise Exception("potential / Hesse expressions may not depend on field velocities")

    plain = C99CodePrinter()._print_Symbol
    names = {x0: "x0", x1: "x1"}
    for sym in set().union(*[e.free_symbols for e in exprs + basis_v]) - {x0, x1}:
        names[sym] = param_slots[plain(sym)]
    functions = []
    for e in exprs:
        if cse is not None:
            repl, red = cse(e)
            functions.append((repl, [red]))
        else:
            functions.append(([], [e]))
    functions.append(cse_vector(basis_v) if cse_vector is not None else ([], basis_v))
    st = Stager(functions, x0, x1, names, staged=staged, regroup=regroup, hoist_reciprocals=hoist_reciprocals, contract_products=contract_products)
    if hoist_reciprocals and share_point_reciprocals:
        # second pass: per-point denominators that serve several quotients get one shared reciprocal
        several = [text for text, n in st.point_den_counts.items() if n >= 2]
        if several:
            st = Stager(functions, x0, x1, names, staged=staged, regroup=regroup, hoist_reciprocals=hoist_reciprocals, shared_point_dens=several, contract_products=contract_products)
    range_flags = st.finish_range_flags()

    idx = {m: {n: k for k, n in enumerate(st.exports[m])} for m in (U, R, C)}

    def imports_for(*users):
        lines, have = [], set()
        for m in (U, R, C):
            for n in st.exports[m]:
                if n not in have and any(u in st.used_by[n] for u in users) and st.stage_of[n] not in users:
                    lines.append(f"  const double {n} = {ARRAY[m]}[{idx[m][n]}];")
                    have.add(n)
        return lines

    def place_imports(imports, statements):
        """Every import (``const double r_7 = R[7];``) goes right before the first statement that reads it.
        Where the tables live in LDS the compiler may not move these loads past the wave-level fences of the
        tile kernel's store path, so a block of imports at the top of the function keeps all of them alive
        over the whole point stage: in source order of first use the tile kernels of the heavy models need
        up to 60 registers less (EGNO: 12 -> 0 spilled registers)."""
        by_name = {}
        for line in imports:
            by_name[line.split()[2]] = line
        placed, out_lines = set(), []
        for line in statements:
            for name in _STAGE_NAME.findall(line):
                if name in by_name and name not in placed:
                    placed.add(name)
                    out_lines.append(by_name[name])
            out_lines.append(line)
        return [by_name[n] for n in by_name if n not in placed] + out_lines

    def body(m):
        lines = list(st.lines[m])
        for n, k in idx[m].items():
            lines.append(f"  {ARRAY[m]}[{k}] = {n};")
        return "\n".join(place_imports(imports_for(m), lines))

    n_par = len(param_slots)
    nu, nr, nc = (len(st.exports[m]) for m in (U, R, C))
    counts = st.statement_counts()
    out = [
        "// Generated by inflatox_amd.Compiler -- do not edit.",
        f"// model: {model_name}; inflatox_amd v{version}; ABI v{abi_version}",
        "#pragma once",
    ]
    for k, v in constants.items():
        out.append(f"#define INFLX_{k} {v}")
    out += [
        f"#define INFLX_N_PARAMETERS {n_par}",
        f"#define INFLX_DIM {model.dim}",
        f'#define INFLX_MODEL_NAME "{model_name}"',
        f"#define INFLX_NU {nu}",
        f"#define INFLX_NR {nr}",
        f"#define INFLX_NC {nc}",
        f"#define INFLX_OUT_MASK {st.out_mask}",
        f"// statements per stage: U={counts[U]} R={counts[R]} C={counts[C]} P={counts[P]}",
        "",
    ]
    tail = "[[maybe_unused]] const double* __restrict__ args"
    out.append("// parameter-only sub-expressions (wave-uniform)")
    out.append(f"INFLX_FN void inflx_stage_uniform({tail}, [[maybe_unused]] double* __restrict__ U) {{")
    out.append(body(U))
    out.append("}\n")
    out.append("// sub-expressions of x[0] (and parameters): once per grid row")
    out.append(
        f"INFLX_FN void inflx_stage_row([[maybe_unused]] const double x0, {tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] double* __restrict__ R) {"
    )
    out.append(body(R))
    out.append("}\n")
    out.append("// sub-expressions of x[1] (and parameters): once per grid column")
    out.append(
        f"INFLX_FN void inflx_stage_col([[maybe_unused]] const double x1, {tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] double* __restrict__ C) {"
    )
    out.append(body(C))
    out.append("}\n")
    point_args = (
        f"[[maybe_unused]] const double x0, [[maybe_unused]] const double x1, {tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] const double* __restrict__ R, "
        "[[maybe_unused]] const double* __restrict__ C, InflxModelValues& mv"
    )
    lines = list(st.lines[P])
    for field, text in zip(OUTPUT_FIELDS, st.outputs):
        lines.append(f"  mv.{field} = {text};")
    if range_flags:
        lines.append("  INFLX_RANGE_CHECK(" + " + ".join(range_flags) + ");")
    point_body = "\n".join(place_imports(imports_for(P, "out"), lines))
    n_inline = point_body.count("INFLX_DIVI(")
    n_hoisted = point_body.count("INFLX_DIVH(") + point_body.count("INFLX_DIVH_PURE(")
    n_shared = point_body.count("INFLX_DIVS(")
    # square roots of the point stage: the quick variant takes them without operand scaling and zero / infinity selection
    # behind one range test each (inflx_sqrt_checked, csrc/inflx_device_math.h)
    if quick_sqrt and not n_inline:
        point_body = _SQRT_CALL.sub("INFLX_SQRT(", point_body)
        point_body = _HPOW_CALL.sub(lambda m: f"INFLX_HPOW({m.group(1)}, ", point_body)
    n_sqrt = point_body.count("INFLX_SQRT(") + point_body.count("INFLX_HPOW(")
    out.append("// everything that depends on both axes, and the five model values")
    if n_inline:
        # ONE point stage.  A quotient whose denominator comes from an earlier stage is formed from that stage's correctly rounded
        # reciprocal (multiply, FMA, FMA) and accepted by one comparison; a wavefront in which some lane's quotient is not a
        # regular case divides those lanes the IEEE way on the spot -- the values are the IEEE program's always.
        out.append(f"// {n_inline} quotients per point by a denominator of an earlier stage, each checking itself (no second copy of the stage)")
        out.append("#define INFLX_HAS_QUICK_POINT 0")
        out.append("#define INFLX_DIVI(a, b, y) inflx_div_by_hoisted_inline((a), (b), (y))")
        out.append(f"INFLX_FN void inflx_stage_point({point_args}) {{")
        out.append(point_body)
        out.append("}")
        out.append("#undef INFLX_DIVI\n")
    elif n_hoisted or n_shared or n_sqrt:
        # The same statements twice.  `quick` forms the quotients whose denominator comes from an earlier
        # stage with inflx_div_by_hoisted (three full-rate instructions instead of an IEEE division) and
        # reports in `ok` whether every one of them was a regular case; `ieee` divides.  A point that is not
        # regular -- NaN or infinite operands, a zero or tiny numerator, overflow, a denormal quotient --
        # is evaluated again by `ieee`, so the values are those of `ieee` always.
        out.append(f"// {n_hoisted} quotients per point by a denominator of an earlier stage")
        out.append("#define INFLX_HAS_QUICK_POINT 1")
        out.append(f"// ({point_body.count('INFLX_DIVH_PURE(')} of them with a numerator made of earlier-stage values only: validity decided per row / column / sweep)")
        out.append("#define INFLX_DIVH(a, b, y) inflx_div_by_hoisted((a), (b), (y), ok)")
        out.append("#define INFLX_DIVH_PURE(a, b, y) inflx_div_by_hoisted_in_range((a), (b), (y))")
        out.append("#define INFLX_RANGE_CHECK(flags) ok = ok && ((flags) == 0.0)")
        out.append(f"// {n_shared} quotients per point that share their per-point denominator with another one ({point_body.count('INFLX_RCPN(')} reciprocals)")
        out.append("#define INFLX_RCPN(b) inflx_shared_reciprocal((b), ok)")
        out.append("#define INFLX_DIVS(a, b, y) inflx_div_by_shared((a), (b), (y), ok)")
        out.append(f"// {n_sqrt} square roots per point without their special-case handling")
        out.append("#define INFLX_SQRT(x) inflx_sqrt_checked((x), ok)")
        out.append("#define INFLX_HPOW(n, x) inflx_hpow_checked<n>((x), ok)")
        out.append(f"INFLX_FN void inflx_stage_point_quick({point_args}, bool& ok) {{")
        out.append(point_body)
        out.append("}")
        out.append("#undef INFLX_DIVH\n#undef INFLX_DIVH_PURE\n#undef INFLX_RANGE_CHECK\n#undef INFLX_RCPN\n#undef INFLX_DIVS\n#undef INFLX_SQRT\n#undef INFLX_HPOW\n")
        out.append("#define INFLX_SQRT(x) sqrt(x)")
        out.append("#define INFLX_HPOW(n, x) inflx_hpow<n>(x)")
        out.append("#define INFLX_DIVH(a, b, y) ((a) / (b))")
        out.append("#define INFLX_RCPN(b) 0.0")
        out.append("#define INFLX_DIVS(a, b, y) ((a) / (b))")
        out.append("#define INFLX_DIVH_PURE(a, b, y) ((a) / (b))")
        out.append("#define INFLX_RANGE_CHECK(flags) (void)(flags)")
        out.append(f"INFLX_FN void inflx_stage_point_ieee({point_args}) {{")
        out.append(point_body)
        out.append("}")
        out.append("#undef INFLX_DIVH\n#undef INFLX_DIVH_PURE\n#undef INFLX_RANGE_CHECK\n#undef INFLX_RCPN\n#undef INFLX_DIVS\n#undef INFLX_SQRT\n#undef INFLX_HPOW\n")
        out.append(f"INFLX_FN void inflx_stage_point({point_args}) {{")
        out.append("  bool ok = true;")
        out.append("  inflx_stage_point_quick(x0, x1, args, U, R, C, mv, ok);")
        out.append("  if (__builtin_expect(!ok, 0)) inflx_stage_point_ieee(x0, x1, args, U, R, C, mv);")
        out.append("}\n")
    else:
        out.append("#define INFLX_HAS_QUICK_POINT 0")
        out.append(f"INFLX_FN void inflx_stage_point({point_args}) {{")
        out.append(point_body)
        out.append("}\n")
    out.append(_emit_basis_point(model, x0, x1, names, param_slots, tail, cse_vector))
    v01_text, v01_is_v10 = _emit_v01_point(model, exprs[2], x0, x1, names, param_slots, tail, cse)
    out.append(v01_text)
    sweep_lines = "\n".join(ln for ln in point_body.splitlines() if not ln.lstrip().startswith(("mv.b0", "mv.b1")))
    info = dict(
        nu=nu, nr=nr, nc=nc, out_mask=st.out_mask, out_masks=list(st.out_masks), statements={str(k): v for k, v in counts.items()},
        inline_quotients=sweep_lines.count("INFLX_DIVI("),
        hoisted_quotients=sweep_lines.count("INFLX_DIVH(") + sweep_lines.count("INFLX_DIVH_PURE("),
        pure_quotients=sweep_lines.count("INFLX_DIVH_PURE("),
        shared_quotients=sweep_lines.count("INFLX_DIVS("),
        shared_reciprocals=sweep_lines.count("INFLX_RCPN("),
        quick_square_roots=sweep_lines.count("INFLX_SQRT(") + sweep_lines.count("INFLX_HPOW("),
        v01_is_v10=bool(v01_is_v10),
    )  # fmt: skip
    return "\n".join(out), info


def _emit_v01_point(model, v10_expr, x0, x1, names, param_slots, tail, cse):
    """The reference's C function ``v01`` (compiler.py:498-504; Hesse2D loads it, hesse_bindings.rs:202-210, and `hesse` /
    `hesse_array` return it, src/lib.rs:384-462) for the INFLX_OP_HESSE kernels behind ``calc_H`` / ``calc_H_array``.

    The projected Hesse matrix is symmetric, and for every example model sympy gives ``hesse_cmp[0][1]`` the very same
    expression tree as ``hesse_cmp[1][0]``: same tree, same printed C, same bits -- then ``INFLX_V01_IS_V10`` is 1 and the
    kernels return the staged v10.  A symbolic stage that leaves the two components in different forms (a simplification
    that timed out on one of them, a hand-made model) makes the reference evaluate two different C functions; then v01 is
    emitted here as a function of its own: staged like the sweep values (identical nodes once, pow chains), all four
    stages inline and evaluated per point -- a helper off the sweep path, exact in the sense of the module docstring."""
    v01 = sympy.sympify(model.hesse_cmp[0][1])
    if v01 == v10_expr:
        return "// v01 is, node for node, the expression of v10 (symmetric projection): the kernels return the staged v10\n#define INFLX_V01_IS_V10 1\n", True
    plain = C99CodePrinter()._print_Symbol
    names = dict(names)
    for sym in v01.free_symbols - {x0, x1}:
        names[sym] = param_slots[plain(sym)]
    if cse is not None:
        repl, red = cse(v01)
        functions = [(repl, [red])]
    else:
        functions = [([], [v01])]
    st = Stager(functions, x0, x1, names, staged=True)
    lines = [ln for m in (U, R, C, P) for ln in st.lines[m]]
    head = (
        "// the reference's C function v01: its own expression, every stage inline (INFLX_OP_HESSE only)\n"
        "#define INFLX_V01_IS_V10 0\n"
        f"INFLX_FN double inflx_v01_point([[maybe_unused]] const double x0, [[maybe_unused]] const double x1, {tail}) {{"
    )
    return head + "\n" + "\n".join(lines) + f"\n  return {st.outputs[0]};\n}}\n", False


def _emit_basis_point(model, x0, x1, names, param_slots, tail, cse_vector):
    """``inflx_basis_point``: the basis vectors ``v``, ``w1`` and their three metric inner products at one
    point -- what ``validate_basis_*`` (reference src/lib.rs:141-300) obtains from the C functions ``v``,
    ``w1`` and ``inner_prod`` (compiler.py:417-472).  Off the sweep path, so it is printed unstaged (one
    self-contained function, evaluated per point) and, like the reference's ``inner_prod``, skips metric
    components that print as zero and sums ``0.0 + (g_ij * v1[i] * v2[j]) + ...`` from left to right."""
    dim = model.dim
    vectors = [[sympy.sympify(c) for c in model.basis[k]] for k in range(dim)]
    metric = [sympy.sympify(model.metric[i][j]) for i in range(dim) for j in range(dim)]
    plain = C99CodePrinter()._print_Symbol
    names = dict(names)
    for sym in set().union(*[e.free_symbols for e in metric + [c for vec in vectors for c in vec]]) - {x0, x1}:
        names[sym] = param_slots[plain(sym)]
    functions = [cse_vector(vec) if cse_vector is not None else ([], vec) for vec in vectors]
    functions.append(cse_vector(metric) if cse_vector is not None else ([], metric))
    # staged like the sweep values -- identical nodes once, pow chains -- with every stage inline, evaluated per point (exact in the
    # sense of the module docstring): printed as the reference's three separate C functions this helper was 200 kB of text for D5 and
    # two thirds of the model's whole hipcc time
    st = Stager(functions, x0, x1, names, staged=True)
    lines = [ln for m in (U, R, C, P) for ln in st.lines[m]]
    texts = list(st.outputs)
    comps = ["bv", "bw"]
    for k in range(dim):
        for i in range(dim):
            lines.append(f"  const double {comps[k]}{i} = {texts[k * dim + i]};")
    gnames = {}
    for i in range(dim):
        for j in range(dim):
            t = texts[dim * dim + i * dim + j]
            if t in ("0", "0.0"):
                continue
            gnames[(i, j)] = f"g{i}{j}"
            lines.append(f"  const double g{i}{j} = {t};")

    def inner(a, b):
        return "0.0" + "".join(f" + (g{i}{j} * {a}{i} * {b}{j})" for (i, j) in gnames)

    lines += [
        f"  o[0] = {inner('bv', 'bv')};",
        f"  o[1] = {inner('bv', 'bw')};",
        f"  o[2] = {inner('bw', 'bw')};",
        "  o[3] = bv0; o[4] = bv1; o[5] = bw0; o[6] = bw1;",
    ]
    head = (
        "// basis vectors v, w1 and their inner products v.v, v.w1, w1.w1 at one point (validate_basis_*)\n"
        f"INFLX_FN void inflx_basis_point([[maybe_unused]] const double x0, [[maybe_unused]] const double x1, {tail}, "
        "double* __restrict__ o) {"
    )
    return head + "\n" + "\n".join(lines) + "\n}\n"
