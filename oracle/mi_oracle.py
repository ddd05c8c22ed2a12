"""CPU restatement of L-GIREMI's site-pair MI step on the reference's own input type.

TEST INFRASTRUCTURE ONLY. Nothing under ``oracle/`` is imported by the product
package (``l-giremi_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it, and only as the checker.

Parity status: PINNED for counts / MI / mean MI — this file is checked against
golden vectors produced by importing the reference itself in the build
container (tests/golden/gen_golden.py -> tests/golden/*.json).  The permutation
p-value has no reference counterpart ("parity unpinned"); its
CPU statement lives in oracle/lgmi_oracle.c.

Reference followed (gxiaolab/L-GIREMI v0.2.4):
  src/giremi/mutual_information.py:6-45    mismatch_pair_mutual_info
  src/giremi/mutual_information.py:48-60   mean_mismatch_pair_mutual_info
  src/giremi/mismatch.py:384-404           caller: per strand, het_snp filter, mean
and the third-party arithmetic it calls (un-vendored dependency, scikit-learn,
unpinned in pyproject.toml:7-12; 1.7.2 installed here):
  sklearn/metrics/cluster/_supervised.py:811-923  mutual_info_score
  sklearn/metrics/cluster/_supervised.py:86       contingency_matrix
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

EPS = 2.220446049250313e-16  # np.finfo(float64).eps, _supervised.py:922


def site_read_classes(site: dict) -> Dict[str, int]:
    """read name -> class (2 major, 1 minor, 0 other) for one site.

    mutual_information.py:15-16: a read listed under several alleles keeps the
    LAST allele in ``nt`` insertion order (dict() construction).
    mutual_information.py:25-40: alleles ranked by the site-wide ``depth`` dict,
    descending, stable (ties keep ``depth`` insertion order); rank 0 -> 2,
    rank 1 -> 1, everything else -> 0 (defaultdict(int)).  A site whose
    ``depth`` holds fewer than two alleles raises IndexError (:30,:32).
    """
    read_allele: Dict[str, str] = {}
    for allele, names in site['nt'].items():
        for name in names:
            read_allele[name] = allele
    ranked = sorted(site['depth'].items(), key=lambda kv: kv[1], reverse=True)
    major = ranked[0][0]
    minor = ranked[1][0]  # IndexError with < 2 alleles, like the reference
    out = {}
    for name, allele in read_allele.items():
        out[name] = 2 if allele == major else (1 if allele == minor else 0)
    return out


def contingency_3x3(c1: Dict[str, int], c2: Dict[str, int]) -> List[List[int]]:
    """3x3 table over the reads present at both sites (mutual_information.py:17)."""
    table = [[0, 0, 0], [0, 0, 0], [0, 0, 0]]
    small, big, swap = (c1, c2, False) if len(c1) <= len(c2) else (c2, c1, True)
    for name, a in small.items():
        b = big.get(name)
        if b is not None:
            if swap:
                table[b][a] += 1
            else:
                table[a][b] += 1
    return table


def mi_from_table(table: Sequence[Sequence[int]]) -> float:
    """sklearn mutual_info_score on a contingency table (_supervised.py:903-923).

    Rows/columns that are empty among the common reads do not exist for sklearn
    (np.unique on the labels); one surviving row or column -> exactly 0.0.
    Natural log, every term with |t| < eps zeroed, sum clipped at 0.
    """
    rows = [sum(r) for r in table]
    cols = [sum(table[a][b] for a in range(len(table))) for b in range(len(table[0]))]
    n = sum(rows)
    if n == 0:
        raise ValueError('math domain error')  # log(0) in the reference path
    if sum(1 for r in rows if r) == 1 or sum(1 for c in cols if c) == 1:
        return 0.0
    log_n = math.log(n)
    total = 0.0
    for a, r in enumerate(rows):
        for b, c in enumerate(cols):
            nab = table[a][b]
            if nab == 0:
                continue
            frac = nab / n
            term = frac * (math.log(nab) - log_n) + frac * (-math.log(r * c) + log_n + log_n)
            if abs(term) < EPS:
                term = 0.0
            total += term
    return max(total, 0.0)


def pair_rows(mismatches: dict, min_common_reads: int = 5, with_counts: bool = False):
    """mutual_information.py:6-45 — rows [p1, type1, p2, type2, mi] for every pair
    of positions (sorted, combinations order) sharing >= min_common_reads reads."""
    positions = sorted(mismatches.keys())
    members = {p: {n for names in mismatches[p]['nt'].values() for n in names} for p in positions}
    classes: Dict[int, Dict[str, int]] = {}

    def cls(p):  # ranked lazily: the reference only ranks alleles of pairs that pass :19
        if p not in classes:
            classes[p] = site_read_classes(mismatches[p])
        return classes[p]

    rows = []
    tables = []
    for a in range(len(positions)):
        for b in range(a + 1, len(positions)):
            p1, p2 = positions[a], positions[b]
            if len(members[p1] & members[p2]) < min_common_reads:
                continue
            table = contingency_3x3(cls(p1), cls(p2))
            rows.append([p1, mismatches[p1]['type'], p2, mismatches[p2]['type'],
                         mi_from_table(table)])
            tables.append(table)
    return (rows, tables) if with_counts else rows


def mean_rows(rows) -> List[List]:
    """mutual_information.py:48-60 — [pos, mean mi] in first-appearance order; each
    row counts for both of its sites; left-to-right float sum."""
    acc: Dict[int, List[float]] = {}
    for p1, _t1, p2, _t2, mi in rows:
        acc.setdefault(p1, []).append(mi)
        acc.setdefault(p2, []).append(mi)
    return [[pos, sum(v) / len(v)] for pos, v in acc.items()]


def region_mi(mismatches_by_strand: dict, min_common_reads: int = 5):
    """mismatch.py:384-404 — per strand ('+' then '-'): all-pairs MI when the strand
    has > 1 site, keep rows with a het_snp side, mean MI over the kept rows."""
    kept = {'+': [], '-': []}
    means = {'+': [], '-': []}
    for strand in ('+', '-'):
        sites = mismatches_by_strand.get(strand, {})
        if len(sites) > 1:
            full = pair_rows(sites, min_common_reads)
            kept[strand] = [r for r in full if r[1] == 'het_snp' or r[3] == 'het_snp']
            if kept[strand]:
                means[strand] = mean_rows(kept[strand])
    return kept, means


def ecdf_strict(sample: Sequence[float]):
    """stat.py:7-29 — f(v) = #{sample < v} / len(sample) (searchsorted side='left')."""
    xs = sorted(sample)
    n = len(xs)
    import bisect

    def f(v):
        return bisect.bisect_left(xs, v) / n
    return f
