# -*- coding: utf-8 -*-
"""Prototype (numpy, CPU) for a TWO-LEVEL prefix: composition of two chunk transfer elements.

The scan's prefix phase walks a problem's chunks one after the other (csrc/clr_batch_kernels.h: prefix_coop_kernel,
64 chunks x ~2.8 us = 0.18 ms of the 2.5 ms headline step, half of BASELINE config 1, half of one long series through
the object API).  The elements (A, b, C, eta, Jm) of csrc/clr_core.h act on the state (P, f) as

    P' = C + A P (I + Jm P)^-1 A^T ,    f' = A (I + P Jm)^-1 (f + P eta) + b ,

and such maps compose in closed form (Sarkka & Garcia-Fernandez, IEEE TAC 66 (2021), Lemma 8 -- element 1 first):

    A12   = A2 (I + C1 Jm2)^-1 A1
    b12   = A2 (I + C1 Jm2)^-1 (b1 + C1 eta2) + b2
    C12   = A2 (I + C1 Jm2)^-1 C1 A2^T + C2
    eta12 = A1^T (I + Jm2 C1)^-1 (eta2 - Jm2 b1) + eta1
    Jm12  = A1^T (I + Jm2 C1)^-1 Jm2 A1 + Jm1 .

With groups of g chunks: compose inside the groups in parallel (g - 1 compositions deep), advance the state over the
G = nchunk / g group elements, then advance inside the groups in parallel (g - 1 deep): depth ~ 2.5 g + G + g
advance-equivalents instead of nchunk (a composition is an advance with 2 J more right-hand sides).  This file is the
algebra only, checked by tests/test_host_api.py::test_element_composition_prototype; nothing in the product uses it.
"""
import numpy as np


def single_step(u, v, phi, a, y):
    """Element of one sample (clr_core.h header): A = Phi (I - v u^T / a), b = Phi v y / a, C = Phi v v^T Phi / a,
    eta = -u y / a, Jm = -u u^T / a."""
    Phi = np.diag(phi)
    J = len(u)
    return (Phi.dot(np.eye(J) - np.outer(v, u) / a), Phi.dot(v) * y / a, Phi.dot(np.outer(v, v)).dot(Phi) / a,
            -u * y / a, -np.outer(u, u) / a)


def advance(elem, P, f):
    A, b, C, eta, Jm = elem
    I = np.eye(len(f))
    Pn = C + A.dot(P).dot(np.linalg.solve(I + Jm.dot(P), A.T))
    fn = A.dot(np.linalg.solve(I + P.dot(Jm), f + P.dot(eta))) + b
    return 0.5 * (Pn + Pn.T), fn


def compose(e1, e2):
    """The element of `e1 then e2`."""
    A1, b1, C1, eta1, J1 = e1
    A2, b2, C2, eta2, J2 = e2
    I = np.eye(len(b1))
    M = np.linalg.inv(I + C1.dot(J2))          # (I + C1 Jm2)^-1
    Mt = np.linalg.inv(I + J2.dot(C1))         # (I + Jm2 C1)^-1 = M^T for symmetric C1, Jm2
    A = A2.dot(M).dot(A1)
    b = A2.dot(M).dot(b1 + C1.dot(eta2)) + b2
    C = A2.dot(M).dot(C1).dot(A2.T) + C2
    eta = A1.T.dot(Mt).dot(eta2 - J2.dot(b1)) + eta1
    Jm = A1.T.dot(Mt).dot(J2).dot(A1) + J1
    return A, b, 0.5 * (C + C.T), eta, 0.5 * (Jm + Jm.T)


def chunk_element(us, vs, phis, a_s, ys):
    """Fold a chunk's samples into one element by composing single-step elements (the kernels use the
    Sherman-Morrison recurrences of clr_core.h instead; same element)."""
    e = single_step(us[0], vs[0], phis[0], a_s[0], ys[0])
    for k in range(1, len(us)):
        e = compose(e, single_step(us[k], vs[k], phis[k], a_s[k], ys[k]))
    return e


def two_level_starts(elems, g):
    """Start states of every chunk from the chunk elements, groups of g chunks: returns the list of (P, f)."""
    J = len(elems[0][1])
    groups = [elems[i:i + g] for i in range(0, len(elems), g)]
    summaries = []
    for grp in groups:                      # level 1: parallel over groups, g - 1 compositions each
        e = grp[0]
        for x in grp[1:]:
            e = compose(e, x)
        summaries.append(e)
    P, f = np.zeros((J, J)), np.zeros(J)
    gstarts = []
    for e in summaries:                     # level 2: sequential over the groups
        gstarts.append((P, f))
        P, f = advance(e, P, f)
    starts = []
    for grp, (P, f) in zip(groups, gstarts):  # level 3: parallel over groups, g - 1 advances each
        for x in grp:
            starts.append((P, f))
            P, f = advance(x, P, f)
    return starts
