#!/usr/bin/env python3
"""Pins lgmi's pile-up semantics to pysam's — the one part of the BAM side that the build container could not check:
pysam / htslib are not installed there, and the reference holds no fixture (src/giremi/mismatch.py:161-188 calls
`sam.pileup(contig, start, stop)` with pysam's defaults: stepper 'samtools', min_base_quality 13, ignore_orphans, max_depth
8000).  Run this WHERE pysam IS INSTALLED:

    python tools/make_pysam_pileup_golden.py            # -> tests/golden/pileup_pysam.json

It writes the deterministic synthetic BAM of tests/helpers/pileup_bam.py with lgmi's own writer, reads it back with pysam and
records, for three regions, every pile-up column exactly as the reference consumes it: pos, get_query_names(),
get_query_sequences().  tests/test_pileup_pysam.py then holds lgmi.io.BamReader.pileup against the file (and is skipped
while the file is absent).  Data only: no reference or pysam code is copied."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'l-giremi_amd'), os.path.join(ROOT, 'tests', 'helpers')]


def main():
    try:
        import pysam
    except ImportError:
        sys.exit('pysam is not installed here: run this where it is (pip install pysam), then commit tests/golden/pileup_pysam.json')
    from pileup_bam import REGIONS, write_bam
    out = {'meta': {'pysam': pysam.__version__, 'call': 'AlignmentFile.pileup(contig, start, stop)  (all defaults)',
                    'bam': 'tests/helpers/pileup_bam.py: write_bam'}, 'regions': []}
    with tempfile.TemporaryDirectory() as d:
        bam = os.path.join(d, 'p.bam')
        write_bam(bam)
        sam = pysam.AlignmentFile(bam)
        for contig, start, stop in REGIONS:
            cols = []
            for col in sam.pileup(contig, start, stop):
                cols.append([col.pos, col.get_query_names(), col.get_query_sequences()])
            out['regions'].append({'contig': contig, 'start': start, 'stop': stop, 'columns': cols})
            print('%s:%d-%d  %d columns' % (contig, start, stop, len(cols)))
    path = os.path.join(ROOT, 'tests', 'golden', 'pileup_pysam.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    print('wrote', path)


if __name__ == '__main__':
    main()
