"""UAI (MARKOV) reader for unary / pairwise models: the text format of the reference's MRF tests
(reference test/graphical_model.cpp:11-30).  Table entries are taken as costs, as the reference's test does;
variables without a unary table get a zero unary ("not all unaries are present, hence zero unaries must be
added", graphical_model.cpp:10).  Builds the model through the LP mirror in variable order with relations
u_i -> p_ij -> u_j (i < j), i.e. what LP_MP-MRF's problem constructor feeds LP<FMC_SRMP>."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import lp as LPM
from . import model as M


def parse_uai(text: str) -> Tuple[List[int], List[Tuple[Tuple[int, ...], np.ndarray]]]:
    tok = text.split()
    if not tok or tok[0] != "MARKOV":
        raise RuntimeError("UAI input must start with MARKOV")
    pos = 1
    n = int(tok[pos]); pos += 1
    card = [int(t) for t in tok[pos:pos + n]]; pos += n
    n_f = int(tok[pos]); pos += 1
    scopes = []
    for _ in range(n_f):
        k = int(tok[pos]); pos += 1
        scopes.append(tuple(int(t) for t in tok[pos:pos + k])); pos += k
    tables = []
    for sc in scopes:
        size = int(tok[pos]); pos += 1
        vals = np.array([float(t) for t in tok[pos:pos + size]]); pos += size
        if size != int(np.prod([card[v] for v in sc])):
            raise RuntimeError("UAI table size does not match its scope")
        tables.append((sc, vals.reshape([card[v] for v in sc])))
    return card, tables


def FMC_SRMP():
    U = LPM.FactorContainer(LPM.UnarySimplexFactor, 0, True)     # COMPUTE_PRIMAL_SOLUTION on the unaries: rounding
    P = LPM.FactorContainer(LPM.PairwiseSimplexFactor, 1)
    ML = LPM.MessageContainer(LPM.UnaryPairwiseMessage(0), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 0)
    MR = LPM.MessageContainer(LPM.UnaryPairwiseMessage(1), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 1)
    return LPM.FMC("FMC_SRMP", [U, P], [ML, MR]), U, P, ML, MR


def build_lp_from_uai(text: str, device: int = 0, order: str = "index") -> LPM.LP:
    """``order``: "index" — relations u_i -> p_ij -> u_j for i < j in the file's variable numbering, what LP_MP-MRF's
    constructor does (a row-major grid then has H+W-1 dependent steps per sweep); "colour_major" — the same relations
    along a colour-major ranking of the variables (ordering.colour_major_order: 2 steps per sweep on a bipartite
    graph).  Both are valid block-coordinate-ascent orders; they give different dual trajectories."""
    card, tables = parse_uai(text)
    if order not in ("index", "colour_major"):
        raise ValueError(order)
    rank = np.arange(len(card))
    if order == "colour_major":
        from .ordering import colour_major_order
        pairs = np.array([sc for sc, _ in tables if len(sc) == 2], np.int64).reshape(-1, 2)
        if pairs.shape[0]:
            rank = colour_major_order(len(card), pairs[:, 0], pairs[:, 1])
    fmc, U, P, ML, MR = FMC_SRMP()
    lp = LPM.LP(fmc, device)
    unary = [np.zeros(c) for c in card]
    for sc, t in tables:
        if len(sc) == 1:
            unary[sc[0]] = unary[sc[0]] + t
        elif len(sc) != 2:
            raise RuntimeError("only unary and pairwise UAI factors are supported")
    u = [lp.add_factor(U, c) for c in unary]
    for sc, t in tables:
        if len(sc) != 2:
            continue
        i, j = sc
        if rank[i] > rank[j]:
            i, j, t = j, i, t.T
        p = lp.add_factor(P, card[i], card[j], np.ascontiguousarray(t))
        lp.add_message(ML, u[i], p)
        lp.add_message(MR, u[j], p)
        lp.AddFactorRelation(u[i], p)
        lp.AddFactorRelation(p, u[j])
    return lp


def solve_uai(text: str, device: int = 0, **visitor_options):
    """MAP estimation for a model in UAI format with the message-passing rounding solver — the reference's
    ``MpRoundingSolver<Solver<LP<FMC_SRMP>, StandardVisitor>>`` + ``UaiMrfInput::ParseString`` (test/graphical_model.cpp:
    47-56).  Returns (lower bound, primal cost, labeling of the variables)."""
    lp = build_lp_from_uai(text, device)
    s = LPM.MpRoundingSolver(lp, LPM.StandardVisitor(**visitor_options))
    s.Solve()
    n = len(parse_uai(text)[0])
    x = None if s.solution_ is None else s.solution_[:n, 0].copy()      # unaries are the first n factors
    return s.lower_bound(), s.primal_cost(), x
