"""The synthetic BAM both sides of the pysam pile-up fixture are made from: tools/make_pysam_pileup_golden.py (run by
someone WITH pysam) and tests/test_pileup_pysam.py (which compares lgmi.io's pile-up with what pysam returned).
Deterministic: the same bytes wherever it is written."""
import numpy as np

REGIONS = [('chr1', 100_000, 101_000), ('chr1', 0, 2_000), ('chr2', 50_000, 50_400)]


def write_bam(path):
    """spliced and unspliced reads with soft clips, insertions, deletions, low base qualities and the flags pysam's
    default stepper drops (unmapped, secondary, QC-fail, duplicate) -> the records written"""
    from lgmi.io import BamWriter
    rng = np.random.default_rng(20250812)
    w = BamWriter(str(path), [('chr1', 400_000), ('chrEmpty', 1000), ('chr2', 200_000)], index=True)
    recs = []
    for contig, ln, n in (('chr1', 400_000, 1500), ('chr2', 200_000, 600)):
        starts = np.sort(np.concatenate([rng.integers(0, ln - 30_000, n - 60), rng.integers(0, 1_500, 30),
                                         rng.integers(100_000 if contig == 'chr1' else 50_000, (100_900 if contig == 'chr1' else 50_300), 30)]))
        for k, st in enumerate(starts.tolist()):
            e1, intron, e2 = int(rng.integers(40, 400)), int(rng.integers(50, 20_000)), int(rng.integers(40, 400))
            cig = [(0, e1), (3, intron), (0, e2)] if k % 3 else [(4, 5), (0, e1), (1, 2), (0, e2), (2, 3), (0, 7)]
            qlen = sum(n_ for op, n_ in cig if op in (0, 1, 4))
            seq = ''.join(rng.choice(list('ACGT'), qlen))
            flag = [0, 0, 0, 256, 1024, 4, 512][k % 7] if k % 40 == 0 else 0
            name = '%s_r%05d' % (contig, k)
            w.write(contig, st, name, bool(k & 1), cig, seq, ':%d' % qlen, quality=rng.integers(5, 41, qlen).astype(np.uint8),
                    flag=flag)
            recs.append(name)
    w.close()
    return recs
